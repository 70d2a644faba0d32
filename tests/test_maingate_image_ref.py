"""CPU: the main-gate image restated in tests/advice_ref.py is a SATISFYING assignment on the oracle's cells -- every row
fulfils the main-gate equation with its fixed row, every lookup input is a row of the (tag, value) table, and halo2's
permuted columns built from it obey the lookup argument's invariants.  (The third-party placement itself is unpinned;
this pins that the restatement is self-consistent, which is what the GPU image is then compared with cell for cell.)"""
import random

import pytest

import advice_ref as AR
from oracle_lib import Oracle

FIELDS = {
    "bn254_fr": 21888242871839275222246405745257275088548364400416034343698204186575808495617,
    "bn254_fq": 21888242871839275222246405745257275088696311157297823662689037894645226208583,
    "pasta_fp": 28948022309329048855892746252171976963363056481941560715954676764349967630337,
    "pasta_fq": 28948022309329048855892746252171976963363056481941647379679742748393362948097,
}


def _case(w, L, seed):
    o = Oracle(w, L)
    rng = random.Random(seed)
    bits = w * L
    n = rng.getrandbits(bits) | (1 << (bits - 1)) | 1
    a, b = rng.randrange(n), rng.randrange(n)
    rc, r, st = o.mul_mod(o.limbs(a), o.limbs(b), o.limbs(n))
    assert rc == 0
    return o, [int(x) for x in o.limbs(a)], [int(x) for x in o.limbs(b)], [int(x) for x in o.limbs(n)], st


@pytest.mark.parametrize("w,L,field", [(64, 32, "bn254_fr"), (32, 16, "pasta_fp"), (64, 4, "bn254_fq"), (32, 128, "pasta_fq")])
def test_image_satisfies_the_main_gate_and_the_lookups(w, L, field):
    P = FIELDS[field]
    o, a, b, n, st = _case(w, L, 1000 * w + L)
    im = AR.mul_mod_image(o.p, a, b, n, st, P)
    C = 2 * L - 1
    nrc = (o.p.carry_nsub + 3) // 4
    assert len(im.rows) == 4 * L + 2 * (C + L * L) + L + 4 + (C - 1) * (23 + nrc) + 23 + 1   # ... + assert_one(eq_bit) :1062
    cfg = AR.LookupConfig(AR.range_lens(w, L, rsa=(w == 64)))
    fixed = [AR.fixed_row(k, w, L, o.p.carry_bits, o.p.carry_sub_bits, o.p.carry_nsub, cfg) for k in im.kinds]
    for ri, (cells, f) in enumerate(zip(im.rows, fixed)):
        e_next = im.rows[ri + 1][4] if ri + 1 < len(im.rows) else 0
        assert AR.gate_residual(cells, e_next, f, P) == 0, (ri, im.kinds[ri])
    table = set(cfg.table())
    usable = len(im.rows) + 37
    inputs = AR.lookup_inputs(im.rows, fixed, usable)
    n_range_rows = 4 * L + (C - 1) * nrc
    for name in AR.ARGS:
        assert all(pair in table for pair in inputs[name]), name
    assert sum(1 for pr in inputs["composition_a"] if pr[0]) == n_range_rows
    has_ov = o.p.carry_bits % o.p.carry_sub_bits != 0
    assert sum(1 for pr in inputs["overflow_a"] if pr[0]) == ((C - 1) if has_ov else 0)


@pytest.mark.parametrize("theta_kind", ["random", "colliding"])
def test_permuted_columns_invariants(theta_kind):
    """halo2's lookup argument needs: A' is a permutation of A, S' of S, and on every row A'[i] == S'[i] or A'[i] == A'[i-1].
    `colliding`: a theta under which two different table rows compress to the same element (theta = 1: 1 * theta + 1 == 2 * theta + 0):
    the algorithm merges them (BTreeMap keyed by the compressed value)."""
    w, L, P = 32, 8, FIELDS["bn254_fr"]
    o, a, b, n, st = _case(w, L, 77)
    im = AR.mul_mod_image(o.p, a, b, n, st, P)
    cfg = AR.LookupConfig(AR.range_lens(w, L))
    fixed = [AR.fixed_row(k, w, L, o.p.carry_bits, o.p.carry_sub_bits, o.p.carry_nsub, cfg) for k in im.kinds]
    usable = 1 << 11
    assert usable >= len(im.rows) and usable >= cfg.n_rows
    theta = 1 if theta_kind == "colliding" else random.Random(5).randrange(P)
    inputs = AR.lookup_inputs(im.rows, fixed, usable)
    tcol = AR.table_column(cfg, theta, usable, P)
    if theta_kind == "colliding":
        assert len(set(tcol)) < cfg.n_rows
    for name in AR.ARGS:
        A = AR.compress(inputs[name], theta, P)
        a_perm, s_perm = AR.permute_expression_pair(A, tcol)
        assert sorted(A) == a_perm and sorted(tcol) == sorted(s_perm)
        assert all(a_perm[i] == s_perm[i] or (i and a_perm[i] == a_perm[i - 1]) for i in range(usable))


@pytest.mark.parametrize("w,L,field,case", [(64, 32, "bn254_fr", "lt"), (32, 16, "pasta_fp", "lt"), (64, 4, "bn254_fq", "ge"), (64, 8, "pasta_fq", "eq")])
def test_in_field_image_satisfies_the_main_gate(w, L, field, case):
    """The rows of assert_in_field(x, n) restated in tests/advice_ref.in_field_image from the oracle's in-field stream: every row
    fulfils the main-gate equation (incl. select / not / assert_one), every range row's cells are table rows; x >= n makes the
    final assert_one row -- and only that row -- unsatisfied."""
    P = FIELDS[field]
    o = Oracle(w, L)
    rng = random.Random(7 * w + L)
    bits = w * L
    n = rng.getrandbits(bits) | (1 << (bits - 1)) | 1
    x = {"lt": rng.randrange(n), "ge": min(n + 3, (1 << bits) - 1), "eq": n}[case]
    rc, lt, st = o.assert_in_field(o.limbs(x), o.limbs(n))
    assert lt == (1 if x < n else 0)
    im = AR.in_field_image(o.p, [int(v) for v in o.limbs(x)], [int(v) for v in o.limbs(n)], st, P)
    cfg = AR.LookupConfig(AR.range_lens(w, L))
    table = set(cfg.table())
    bad = []
    for ri, (cells, kind) in enumerate(zip(im.rows, im.kinds)):
        f = AR.fixed_row(kind, w, L, o.p.carry_bits, o.p.carry_sub_bits, o.p.carry_nsub, cfg)
        e_next = im.rows[ri + 1][4] if ri + 1 < len(im.rows) else 0
        if AR.gate_residual(cells, e_next, f, P) != 0:
            bad.append(ri)
        if f["tag_composition"]:
            assert all((f["tag_composition"], cells[c]) in table for c in range(4)), ri
    assert bad == ([] if x < n else [len(im.rows) - 1]), bad[:5]
    assert sum(1 for k in im.kinds if k == AR.ROW_RANGE_LIMB) == 8 * L + 6      # the range assigns h2r_lookup_hist_fresh_op counts


@pytest.mark.parametrize("w,L,field", [(64, 8, "bn254_fr"), (32, 16, "pasta_fp")])
def test_fresh_family_images_satisfy_the_main_gate(w, L, field):
    """Every op of the Fresh-integer family restated as rows (tests/advice_ref.fresh_image) from the oracle's stream of the op."""
    from oracle_lib import FRESH_OPS, fresh_op
    P = FIELDS[field]
    o = Oracle(w, L)
    rng = random.Random(w + L)
    bits = w * L
    n = rng.getrandbits(bits) | (1 << (bits - 1)) | 1
    cfg = AR.LookupConfig(AR.range_lens(w, L))
    fixed = {}
    for a, b in [(rng.randrange(n), rng.randrange(n)), (5, 5), (0, n - 1), (n - 1, 0)]:
        for name in FRESH_OPS:
            rc, ov, of, st = fresh_op(o, name, o.limbs(a), o.limbs(b), o.limbs(n))
            if rc != 0:
                continue
            im = AR.fresh_image(o.p, name, o.limbs(a), None if name == "is_zero" else o.limbs(b),
                                o.limbs(n) if name in ("add_mod", "sub_mod") else None, st, P)
            for ri, (cells, kind) in enumerate(zip(im.rows, im.kinds)):
                if kind not in fixed:
                    fixed[kind] = AR.fixed_row(kind, w, L, o.p.carry_bits, o.p.carry_sub_bits, o.p.carry_nsub, cfg)
                e_next = im.rows[ri + 1][4] if ri + 1 < len(im.rows) else 0
                assert AR.gate_residual(cells, e_next, fixed[kind], P) == 0, (name, ri, kind)


def test_em_check_image_satisfies_the_main_gate():
    """The encoded-message check of verify_pkcs1v15_signature (src/chip.rs:138-198) restated as rows (advice_ref.em_image) from
    the oracle's EM stream, for KAT 1 (valid) and a tampered digest: every row fulfils the gate, the 4-bit sub-limbs of the two
    32-bit range assigns are rows of RSAChip's table; the final running AND is is_valid."""
    import json
    import os
    P = FIELDS["bn254_fr"]
    o = Oracle(64, 32)
    k = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "halo2_rsa_golden.json")))["rsa_kats"][0]
    n, sig, hashed = int(k["n"]), int(k["sig"]), int(k["hashed"])
    powed = o.limbs(pow(sig, 65537, n))
    cfg = AR.LookupConfig(AR.range_lens(64, 32, rsa=True))
    table = set(cfg.table())
    for tamper in (0, 1):
        h4 = o.limbs(hashed ^ tamper, 4)
        rc, valid, st = o.pkcs1v15_em_check(powed, h4)
        im, is_eq = AR.em_image(o.p, powed, h4, st, P)
        assert is_eq == valid == (1 - tamper)
        assert len(im.rows) == 178      # 33 comparisons x 5 rows, 7 constants, 4 range rows, mul_add, assert_equal
        for ri, (cells, kind) in enumerate(zip(im.rows, im.kinds)):
            f = AR.fixed_row(kind, 64, 32, o.p.carry_bits, o.p.carry_sub_bits, o.p.carry_nsub, cfg)
            e_next = im.rows[ri + 1][4] if ri + 1 < len(im.rows) else 0
            assert AR.gate_residual(cells, e_next, f, P) == 0, (ri, kind)
            if f["tag_composition"]:
                assert all((f["tag_composition"], cells[c]) in table for c in range(4)), ri


@pytest.mark.parametrize("w,L,field,e_limbs,nb", [(64, 4, "bn254_fr", [0b10110, 0b00001, 0b11111, 0], 5), (32, 8, "pasta_fq", [0x5A5A5A5A, 1], 32),
                                                   (64, 4, "bn254_fq", [0x8000000000000001], 64), (32, 8, "pasta_fp", [0b1011011, 0b0000001, 0b1111111], 7)])
def test_pow_var_image_satisfies_the_main_gate(w, L, field, e_limbs, nb):
    """BigIntChip::pow_mod (big_integer/chip.rs:664-696) restated as rows: main_gate.to_bits of every exponent limb (bits, their
    composition rows, assert_equal), acc = 1, then per bit mul_mod / select / square_mod -- every row satisfies the main gate with its
    fixed row, the composition rows carry no lookup, and the SELECT rows hold select(muled, acc, e_bit)."""
    P = FIELDS[field]
    o = Oracle(w, L)
    rng = random.Random(w + L + nb)
    bits = w * L
    n = rng.getrandbits(bits) | (1 << (bits - 1)) | 1
    x = rng.randrange(n)
    rc, out, st = o.pow_mod(o.limbs(x), e_limbs, nb, o.limbs(n))
    assert rc == 0
    e = sum(v << (nb * i) for i, v in enumerate(e_limbs))
    assert sum(int(v) << (w * i) for i, v in enumerate(out)) == pow(x, e, n)
    im = AR.pow_var_image(o.p, [int(v) for v in o.limbs(x)], e_limbs, nb, [int(v) for v in o.limbs(n)], st, P, o.mul_mod_stream_bytes)
    nbits = len(e_limbs) * nb
    per_limb = nb + (nb + 3) // 4 + 1
    C = 2 * L - 1
    nrc = (o.p.carry_nsub + 3) // 4
    rows_mm = 4 * L + 2 * (C + L * L) + L + 4 + (C - 1) * (23 + nrc) + 23 + 1   # ... + assert_one(eq_bit) :1062
    assert len(im.rows) == len(e_limbs) * per_limb + 2 + nbits * (2 * rows_mm + L)
    cfg = AR.LookupConfig(AR.range_lens(w, L, rsa=(w == 64)))

    def fixed_of(k):
        if AR.ROW_BITS_COMPOSE <= k < AR.ROW_BITS_COMPOSE_LAST + 64:
            return AR.fixed_row_bits_compose(k)
        return AR.fixed_row(k, w, L, o.p.carry_bits, o.p.carry_sub_bits, o.p.carry_nsub, cfg)
    cache = {}
    for ri, (cells, k) in enumerate(zip(im.rows, im.kinds)):
        f = cache.setdefault(k, fixed_of(k))
        e_next = im.rows[ri + 1][4] if ri + 1 < len(im.rows) else 0
        assert AR.gate_residual(cells, e_next, f, P) == 0, (ri, k)
    # the first limb's to_bits rows, spelled out
    lb = e_limbs[0]
    assert [r[0] for r in im.rows[:nb]] == [(lb >> t) & 1 for t in range(nb)] and im.kinds[:nb] == [AR.ROW_BIT] * nb
    assert im.rows[nb][4] == lb and im.rows[per_limb - 1][:2] == [lb, lb] and im.kinds[per_limb - 1] == AR.ROW_ASSERT_EQ
    assert im.kinds.count(AR.ROW_SELECT) == nbits * L
