"""Pins the oracle: C restatement (oracle/h2r_oracle.c) vs the reference's known-answer vectors
(tests/golden/halo2_rsa_golden.json, minted by tests/golden/make_golden.py) and vs the independent
Python big-int restatement (oracle/pyref.py).  CPU only."""
import hashlib
import os
import random
import sys

import numpy as np
import pytest

from oracle_lib import Oracle

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
import pyref as R  # noqa: E402


def sha(a):
    return hashlib.sha256(bytes(a)).hexdigest()


def test_rsa_kats_pow_stream_and_validity(golden):
    """reference src/chip.rs:683-816: KAT1, KAT2 valid; BAD invalid."""
    o = Oracle(64, 32)
    for k in golden["rsa_kats"]:
        n, sig, hashed = int(k["n"]), int(k["sig"]), int(k["hashed"])
        rc, out, st = o.pow_mod_fixed_exp(o.limbs(sig), o.limbs(n), k["e"])
        assert rc == 0
        assert len(st) == k["pow_stream_bytes"] == 19 * 64338 + 256
        assert sha(st) == k["pow_stream_sha256"]
        assert ["%x" % int(v) for v in out] == k["powed_limbs"]
        assert o.to_int(out) == pow(sig, k["e"], n)
        rc, ok, em = o.pkcs1v15_em_check(out, o.limbs(hashed, 4))
        assert rc == 0 and ok == k["is_valid"]
        assert len(em) == k["em_stream_bytes"] and sha(em) == k["em_stream_sha256"]
        rc, lt, inf = o.assert_in_field(o.limbs(sig), o.limbs(n))
        assert rc == 0 and lt == 1
        assert len(inf) == k["in_field_stream_bytes"] and sha(inf) == k["in_field_stream_sha256"]
        rc, bp = o.big_pow_mod(o.limbs(sig), k["e"], o.limbs(n))
        assert rc == 0 and np.array_equal(bp, out)


def test_rsa_kat1_every_mul_mod(golden):
    """All 19 (q, r, stream) of KAT1 (17 squarings + 2 multiplies, reference order)."""
    o = Oracle(64, 32)
    k = golden["rsa_kats"][0]
    n = o.limbs(int(k["n"]))
    acc, squared = o.limbs(1), o.limbs(int(k["sig"]))
    ops = [m["op"] for m in k["mul_mods"]]
    assert ops == ["square", "mul"] + ["square"] * 15 + ["square", "mul"]
    it = iter(k["mul_mods"])
    for bit in R.fixed_exp_bits(k["e"]):
        cur = squared
        m = next(it)
        rc, squared, st = o.mul_mod(cur, cur, n)
        assert rc == 0 and sha(st) == m["sha256"] and ["%x" % int(v) for v in squared] == m["r"]
        if bit:
            m = next(it)
            rc, acc, st = o.mul_mod(acc, cur, n)
            assert rc == 0 and sha(st) == m["sha256"] and ["%x" % int(v) for v in acc] == m["r"]


def test_mul_columns_reference_cases(golden):
    """reference big_integer/chip.rs:2797-3107 (un-carried column sums incl. the 16-limb case5)."""
    o = Oracle(64, 32)
    for c in golden["mul_cases"]:
        a = [int(x, 16) for x in c["a"]] + [0] * (32 - len(c["a"]))
        b = [int(x, 16) for x in c["b"]] + [0] * (32 - len(c["b"]))
        cols = o.mul_columns(np.array(a, np.uint64), np.array(b, np.uint64))
        want = [int(x) for x in c["cols"]]
        assert cols[:len(want)] == want, c["name"]
        assert all(v == 0 for v in cols[len(want):])


def test_mul_mod_identities(golden):
    """reference big_integer/chip.rs:3123, 3164, 3204, 3246."""
    o = Oracle(64, 32)
    for c in golden["mul_mod_identities"]:
        rc, r, st = o.mul_mod(o.limbs(int(c["a"])), o.limbs(int(c["b"])), o.limbs(int(c["n"])))
        assert rc == 0 and o.to_int(r) == int(c["r"]) and sha(st) == c["stream_sha256"], c["name"]


def test_parameter_goldens(golden):
    """SURVEY 8 parameter table; big_integer/chip.rs:1220-1249; mod.rs:509."""
    import ctypes
    from oracle_lib import lib
    for row in golden["params"]:
        o = Oracle(row["w"], row["L"])
        assert (o.p.LB, o.p.WB, o.p.CB, o.p.carry_bits, o.p.word_max_bits, o.mul_mod_stream_bytes) == \
            (row["LB"], row["WB"], row["CB"], row["carry_bits"], row["word_max_bits"], row["mul_mod_stream_bytes"])
        comp, over = (ctypes.c_uint32 * 3)(), (ctypes.c_uint32 * 3)()
        lib().h2ro_compute_range_lens(row["w"], row["L"], comp, over)
        assert list(comp) == row["comp"] and list(over) == row["over"]
        assert R.compute_range_lens(row["w"], row["L"]) == (row["comp"], row["over"])
    assert golden["rsa_range_lens_2048"] == [[8, 1, 8, 4], [0, 0, 6]]
    assert R.refresh_aux_increased_limbs(32, 1, 1) == golden["refresh_aux_32_1_1"] == [1, 0]


def test_pow_var_golden(golden):
    """reference big_integer/chip.rs:664-696 with 5-bit exponents (src/chip.rs:283, 327)."""
    o = Oracle(64, 32)
    k = golden["rsa_kats"][0]
    for c in golden["pow_var_kat1"]:
        rc, out, st = o.pow_mod(o.limbs(int(k["sig"])), np.array([c["e"]], np.uint64), c["exp_limb_bits"], o.limbs(int(k["n"])))
        assert rc == 0 and len(st) == c["stream_bytes"] and sha(st) == c["stream_sha256"]
        assert ["%x" % int(v) for v in out] == c["result_limbs"]


@pytest.mark.parametrize("w,L,e_num_limbs,exp_limb_bits", [(64, 4, 2, 33), (64, 4, 2, 64), (64, 8, 32, 3), (32, 8, 2, 32), (32, 8, 5, 7),
                                                          (64, 4, 32, 64)])
def test_pow_var_multi_limb_c_oracle_equals_python(w, L, e_num_limbs, exp_limb_bits):
    """pow_mod with several exponent limbs / wide exp_limb_bits (big_integer/chip.rs:664-696): main_gate.to_bits runs per
    limb, LSB first (:674-681), so e = sum limb_k * 2^(k * exp_limb_bits); both restatements agree and match pow()."""
    o, p = Oracle(w, L), R.Params(w, L)
    rng = random.Random(77 * w + L + exp_limb_bits)
    bits = w * L
    n = rng.getrandbits(bits) | (1 << (bits - 1))
    x = rng.randrange(n)
    e_limbs = [rng.getrandbits(exp_limb_bits) for _ in range(e_num_limbs)]
    e_limbs[0] |= 1 << (exp_limb_bits - 1)
    st = R.Stream()
    r = R.pow_mod_var(p, R.to_limbs(x, L, w), e_limbs, R.to_limbs(n, L, w), exp_limb_bits, st)
    rc, out, cs = o.pow_mod(o.limbs(x), np.array(e_limbs, dtype=o.dtype), exp_limb_bits, o.limbs(n))
    assert rc == 0 and bytes(cs) == st.bytes() and [int(v) for v in out] == r
    e = sum(v << (exp_limb_bits * k) for k, v in enumerate(e_limbs))
    assert o.to_int(out) == pow(x, e, n)
    assert len(cs) == o.pow_var_stream_bytes(e_num_limbs, exp_limb_bits)
    if exp_limb_bits < w:   # a limb that does not fit exp_limb_bits bits cannot satisfy to_bits (:677)
        bad = list(e_limbs); bad[-1] = 1 << exp_limb_bits
        assert o.pow_mod(o.limbs(x), np.array(bad, dtype=o.dtype), exp_limb_bits, o.limbs(n))[0] == 1
        with pytest.raises(ValueError):
            R.pow_mod_var(p, R.to_limbs(x, L, w), bad, R.to_limbs(n, L, w), exp_limb_bits, R.Stream())


def test_rsa4096_w32_golden(golden):
    """BASELINE config 4: RSA-4096 as 128 x 32-bit limbs."""
    o = Oracle(32, 128)
    c = golden["rsa4096_w32"]
    rc, out, st = o.pow_mod_fixed_exp(o.limbs(int(c["x"])), o.limbs(int(c["n"])), c["e"])
    assert rc == 0 and o.to_int(out) == int(c["result"])
    assert len(st) == c["stream_bytes"] == 19 * 563052 + 512 and sha(st) == c["stream_sha256"]


@pytest.mark.parametrize("w,L", [(64, 4), (64, 16), (64, 32), (32, 8), (32, 64), (64, 64), (64, 48), (64, 24), (64, 12), (32, 96), (32, 24)])
def test_c_oracle_equals_python_restatement_random(w, L):
    """Two independent restatements (C: schoolbook + Knuth D; Python: built-in big ints) agree byte for byte.
    Like the reference's random tests (big_integer/chip.rs:1439-1444) n has its top bit set and is not
    forced odd; also small / even moduli."""
    o, p = Oracle(w, L), R.Params(w, L)
    rng = random.Random(1000 * w + L)
    bits = w * L
    for trial in range(6):
        n = rng.getrandbits(bits) | (1 << (bits - 1))
        if trial == 1:
            n = rng.getrandbits(bits // 2 + 3) | 1
        if trial == 2:
            n &= ~1
        a, b = rng.randrange(n), rng.randrange(n)
        if trial == 3:
            a = b = n - 1
        if trial == 4:
            a, b = 0, 1
        st = R.Stream()
        r = R.mul_mod(p, R.to_limbs(a, L, w), R.to_limbs(b, L, w), R.to_limbs(n, L, w), st)
        rc, rr, cs = o.mul_mod(o.limbs(a), o.limbs(b), o.limbs(n))
        assert rc == 0 and bytes(cs) == st.bytes() and [int(v) for v in rr] == r
        assert o.to_int(rr) == (a * b) % n


def test_error_statuses():
    """reference panics: big_integer/chip.rs:566 (n = 0), :583-584 (quotient overflow)."""
    o = Oracle(64, 4)
    one = o.limbs(1)
    rc, _, _ = o.mul_mod(one, one, o.limbs(0))
    assert rc == 2
    big = o.limbs((1 << 256) - 1)
    rc, _, _ = o.mul_mod(big, big, o.limbs(5))
    assert rc == 3
    rc, lt, _ = o.assert_in_field(o.limbs(7), o.limbs(7))
    assert rc == 8 and lt == 0


def test_batch_driver_matches_single():
    o = Oracle(64, 8)
    rng = random.Random(5)
    ns = [rng.getrandbits(512) | (1 << 511) | 1 for _ in range(5)]
    xs = [rng.randrange(n) for n in ns]
    x = np.stack([o.limbs(v) for v in xs]); n = np.stack([o.limbs(v) for v in ns])
    out, status, st = o.pow_mod_fixed_exp_batch(x, n, 65537, nthreads=3, want_stream=True)
    assert not status.any()
    for i in range(5):
        rc, o1, s1 = o.pow_mod_fixed_exp(x[i], n[i], 65537)
        assert np.array_equal(o1, out[i]) and np.array_equal(s1, st[i])
        assert o.to_int(out[i]) == pow(xs[i], 65537, ns[i])


@pytest.mark.parametrize("w,L", [(64, 4), (64, 32), (32, 8), (32, 128)])
def test_fresh_family_c_oracle_equals_python(w, L):
    """add / sub / add_mod / sub_mod / comparisons -- C restatement == Python restatement,
    and the predicates / values agree with plain integer arithmetic (incl. the reference's a+b == n quirk)."""
    from oracle_lib import FRESH_OPS, fresh_op
    assert list(R.FRESH_OPS) == FRESH_OPS
    o, p = Oracle(w, L), R.Params(w, L)
    rng = random.Random(77 * w + L)
    bits = w * L
    n = rng.getrandbits(bits) | (1 << (bits - 1))
    for t in range(8):
        a, b = rng.randrange(n), rng.randrange(n)
        if t == 0:
            b = n - a
        if t == 1:
            b = a
        if t == 2:
            a = 0
        for name in FRESH_OPS:
            st = R.Stream()
            val, fl = R.fresh_op(p, name, R.to_limbs(a, L, w), R.to_limbs(b, L, w), R.to_limbs(n, L, w), st)
            rc, ov, of, ost = fresh_op(o, name, o.limbs(a), o.limbs(b), o.limbs(n))
            assert rc == 0 and bytes(ost) == st.bytes(), (name, t)
            if val is not None:
                assert [int(x) for x in ov] == val
            if fl is not None:
                assert of == fl
        assert R.from_limbs(R.fresh_op(p, "add", R.to_limbs(a, L, w), R.to_limbs(b, L, w), None, R.Stream())[0], w) == a + b
        assert R.fresh_op(p, "is_less_than", R.to_limbs(a, L, w), R.to_limbs(b, L, w), None, R.Stream())[1] == int(a < b)


@pytest.mark.parametrize("w,L", [(64, 4), (64, 32), (32, 128)])
def test_refresh_and_is_equal_muled_c_oracle_equals_python(w, L):
    """refresh (chip.rs:168-233, RefreshAux mod.rs:428-482) and stand-alone is_equal_muled."""
    from oracle_lib import is_equal_muled, mul_stream, refresh
    o, p = Oracle(w, L), R.Params(w, L)
    rng = random.Random(w * L + 1)
    for t in range(3):
        a = [rng.getrandbits(w) for _ in range(L)]
        b = [rng.getrandbits(w) for _ in range(L)]
        if t == 0:
            a = b = [(1 << w) - 1] * L
        cols = R.mul_columns(a, b, None, p.WB)
        ocols, _ = mul_stream(o, np.array(a, o.dtype), np.array(b, o.dtype))
        assert ocols == cols
        st = R.Stream()
        r = R.refresh(p, cols, st)
        rc, rl, rst = refresh(o, cols)
        assert rc == 0 and bytes(rst) == st.bytes() and [int(x) for x in rl] == r
        assert R.from_limbs(r, w) == R.from_limbs(a, w) * R.from_limbs(b, w)
        other = [c + (1 if i == 1 else 0) for i, c in enumerate(cols)]
        for x, want in ((R.mul_columns(b, a, None, p.WB), 1), (other, 0)):
            st = R.Stream()
            assert R.is_equal_muled(p, cols, x, st) == want
            e, est = is_equal_muled(o, cols, x)
            assert e == want and bytes(est) == st.bytes()
