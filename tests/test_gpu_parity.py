"""GPU parity: libh2r.so (hand-written HIP, through the C ABI) vs the CPU oracle, bit-exact.

Every test here needs a real MI355X (`-m gpu`).  The op-trace of each element is walked with
h2r_trace_flatten (the order a layouter shim assigns cells in) and compared byte for byte with the
oracle's stream for the same inputs.  Nothing here reads /root/reference.
"""
import ctypes
import hashlib
import random

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")

from oracle_lib import Oracle  # noqa: E402


def sha(a):
    return hashlib.sha256(bytes(a)).hexdigest()


@pytest.fixture(scope="module")
def H():
    if not torch.cuda.is_available():
        pytest.fail("GPU test selected but no GPU is visible")
    import halo2_rsa_amd
    return halo2_rsa_amd


def rand_modulus(rng, bits, odd=True):
    n = rng.getrandbits(bits) | (1 << (bits - 1))
    return n | 1 if odd else n


def test_rsa_kats_modpow_public_key(H, golden):
    """reference src/chip.rs:683-816 through RSAChip::modpow_public_key (Fix e = 65537)."""
    rsa = H.RSAChip(2048, 5)
    chip = rsa.bigint_chip()
    kats = golden["rsa_kats"]
    ns = [int(k["n"]) for k in kats]
    sigs = [int(k["sig"]) for k in kats]
    pk = rsa.assign_public_key(H.RSAPublicKey(H.UnassignedInteger.from_ints(ns, 32, 64), H.Fix(65537)))
    sig = rsa.assign_signature(H.RSASignature(H.UnassignedInteger.from_ints(sigs, 32, 64)))
    res = rsa.modpow_public_key(sig.c, pk)
    torch.cuda.synchronize()
    assert res.status.cpu().tolist() == [0, 0, 0]
    powed = res.value.to_big_uint()
    o = Oracle(64, 32)
    for i, k in enumerate(kats):
        assert powed[i] == pow(sigs[i], 65537, ns[i])
        assert ["%x" % int(v) for v in res.value.limbs_host()[i]] == k["powed_limbs"]
        st = res.trace.flatten(i)
        assert len(st) == k["pow_stream_bytes"] == 19 * 64338 + 256
        assert sha(st) == k["pow_stream_sha256"], k["name"]
        rc, out, ost = o.pow_mod_fixed_exp(o.limbs(sigs[i]), o.limbs(ns[i]), 65537)
        assert rc == 0 and np.array_equal(ost, st)
        # the whole modpow_public_key witness in the reference's order: assert_in_field (src/chip.rs:106), then the pow path
        rc, lt, s_if = o.assert_in_field(o.limbs(sigs[i]), o.limbs(ns[i]))
        assert rc == 0 and lt == 1
        assert np.array_equal(res.in_field.flatten(i), s_if) and sha(s_if) == k["in_field_stream_sha256"]
        assert np.array_equal(res.flatten(i), np.concatenate([s_if, ost]))
        # EM check of the pkcs1v15 verifier on the GPU result (expected is_valid = 1, 1, 0)
        rc, ok, _ = o.pkcs1v15_em_check(res.value.limbs_host()[i], o.limbs(int(k["hashed"]), 4))
        assert ok == k["is_valid"]


@pytest.mark.parametrize("w,L,batch", [(64, 32, 48), (64, 16, 24), (32, 128, 6), (64, 64, 6), (32, 64, 8), (64, 4, 40), (32, 8, 40),
                                       # num_limbs that are not powers of two (BigIntChip::new only asserts bits_len % limb_width == 0,
                                       # big_integer/chip.rs:1174-1185): RSA-3072 (L = 48), RSA-1536 (24), 768 bits, odd multiples of 4 / 8
                                       (64, 48, 24), (64, 24, 24), (64, 12, 40), (64, 20, 16), (64, 60, 8), (32, 96, 8), (32, 24, 16),
                                       (32, 72, 8), (32, 120, 6),
                                       # 65..96 digits: the chain kernel's K = 96 build (partial last lane group), exact and padded
                                       (64, 40, 16), (64, 44, 16), (32, 88, 8), (64, 36, 16)])
def test_mul_mod_random_parity(H, w, L, batch):
    """mul_mod (reference big_integer/chip.rs:542-629) incl. the reference's own edge identities
    (:3123-3246), even and small moduli (the reference's random n is not forced odd, :1439-1442)."""
    chip = H.BigIntChip(w, w * L)
    o = Oracle(w, L)
    rng = random.Random(7 * w + L)
    bits = w * L
    A, B, N = [], [], []
    for i in range(batch):
        kind = i % 8
        n = rand_modulus(rng, bits, odd=(kind != 1))
        if kind == 2:
            n = rng.getrandbits(bits // 2 + 5) | 1          # small modulus: leading zero limbs
        if kind == 3:
            n = rng.getrandbits(bits - 7) | (1 << (bits - 8))  # top bit clear: normalisation shift
        a, b = rng.randrange(n), rng.randrange(n)
        if kind == 4:
            a, b = 0, rng.randrange(n)                       # 0 * b = 0
        if kind == 5:
            a, b = n - 1, n - 1                               # (n-1)^2 = 1
        if kind == 6:
            a, b = n - 1, max(n - 2, 0)                       # (n-1)(n-2) = 2
        if kind == 7:
            a, b = n, 1                                       # n * 1 = 0 (a == n still has q fitting)
        A.append(a); B.append(b); N.append(n)
    res = chip.mul_mod(chip.assign_integer(A), chip.assign_integer(B), chip.assign_integer(N))
    torch.cuda.synchronize()
    assert not res.status.cpu().numpy().any()
    r = res.value.to_big_uint()
    for i in range(batch):
        assert r[i] == (A[i] * B[i]) % N[i], (i, i % 8)
        rc, rr, ost = o.mul_mod(o.limbs(A[i]), o.limbs(B[i]), o.limbs(N[i]))
        assert rc == 0
        st = res.trace.flatten(i)
        if not np.array_equal(ost, st):
            bad = int(np.nonzero(ost != st)[0][0])
            pytest.fail("w=%d L=%d elem %d kind %d: first stream mismatch at byte %d of %d" % (w, L, i, i % 8, bad, len(st)))


def _adversarial_cases(bits, rng):
    """(a, b, n) triples that stress carry propagation across the 32-bit digits, the 64-lane groups and the waves:
    all-ones runs, digits at the extremes, moduli just above a power of two / just below 2^bits / with long 0xff.. or
    zero stretches, operands n-1, and quotient-estimate corner cases (top digits equal)."""
    full = (1 << bits) - 1
    half = bits // 2
    mods = [full, full - 2, (1 << (bits - 1)) + 1, (1 << (bits - 1)) | 1 | (((1 << half) - 1) << 16), (1 << (bits - 1)) + (1 << 32) - 1,
            full ^ (((1 << 64) - 1) << (bits // 2)), (1 << (bits - 1)) | (1 << 31) | 1, full - (1 << (bits - 33)),
            (0x80000000 << (bits - 32)) | ((1 << (bits - 32)) - 1), (0xffffffff << (bits - 32)) | 1,
            (1 << 2047 | 0xffffffffffffffff0000000000000001) if bits == 2048 else (1 << (bits - 1)) | 0xff01]
    out = []
    for n in mods:
        n |= 1 << (bits - 1)
        cands = [n - 1, n - 2, (1 << (bits - 1)) - 1, (1 << half) - 1, ((1 << half) - 1) << (half - 3), full & (n - 1), 1, 0,
                 int("ffffffff00000000" * (bits // 64), 16) % n, int("00000000ffffffff" * (bits // 64), 16) % n, rng.randrange(n)]
        for a in cands[:6]:
            for b in (cands[0], cands[2], cands[8], cands[9], cands[10]):
                out.append((a % n, b % n, n))
    return out


@pytest.mark.parametrize("w,L", [(64, 32), (32, 128), (64, 16), (64, 64), (64, 48), (64, 24), (32, 96), (64, 40)])
def test_mul_mod_adversarial_operands(H, w, L):
    """The chain kernel's ballot carries, DPP neighbour exchange and correction loop on adversarial digits: results
    against Python integers for every triple, full traces against the oracle for a sample.  330 elements keep the
    batch in the latency build's range (<= 512) for RSA-2048; a second, padded batch of 1,100 runs the throughput build."""
    bits = w * L
    chip = H.BigIntChip(w, bits)
    o = Oracle(w, L)
    rng = random.Random(bits + w)
    cases = _adversarial_cases(bits, rng)
    for reps in (1, 4) if (w, L) == (64, 32) else (1,):
        A = [c[0] for c in cases] * reps; B = [c[1] for c in cases] * reps; N = [c[2] for c in cases] * reps
        res = chip.mul_mod(chip.assign_integer(A), chip.assign_integer(B), chip.assign_integer(N))
        torch.cuda.synchronize()
        assert not res.status.cpu().numpy().any()
        r = res.value.to_big_uint()
        bad = [i for i in range(len(A)) if r[i] != (A[i] * B[i]) % N[i]]
        assert not bad, (w, L, reps, bad[:5])
        for i in rng.sample(range(len(A)), 12):
            rc, rr, ost = o.mul_mod(o.limbs(A[i]), o.limbs(B[i]), o.limbs(N[i]))
            assert rc == 0 and np.array_equal(ost, res.trace.flatten(i)), (w, L, i)
    # the same moduli through a long square-and-multiply chain (errors would compound)
    e = (1 << 17) - 1
    n_list = sorted({c[2] for c in cases})
    X = [n - 2 for n in n_list]
    pres = chip.pow_mod_fixed_exp(chip.assign_integer(X), e, chip.assign_integer(n_list), want_trace=False)
    torch.cuda.synchronize()
    assert not pres.status.cpu().numpy().any()
    got = pres.value.to_big_uint()
    assert all(got[i] == pow(X[i], e, n_list[i]) for i in range(len(X)))


def test_mul_mod_golden_identities(H, golden):
    chip = H.BigIntChip(64, 2048)
    ids = golden["mul_mod_identities"]
    res = chip.mul_mod(chip.assign_integer([int(c["a"]) for c in ids]), chip.assign_integer([int(c["b"]) for c in ids]),
                       chip.assign_integer([int(c["n"]) for c in ids]))
    torch.cuda.synchronize()
    for i, c in enumerate(ids):
        assert res.value.to_big_uint()[i] == int(c["r"])
        assert sha(res.trace.flatten(i)) == c["stream_sha256"], c["name"]


def test_error_statuses(H):
    """n = 0 (reference divides by zero, chip.rs:566); quotient overflow (:583-584); x >= n (src/chip.rs:106)."""
    chip = H.BigIntChip(64, 2048)
    big = (1 << 2048) - 1
    n_ok = (1 << 2047) | 12345
    res = chip.mul_mod(chip.assign_integer([3, big, 5, big]), chip.assign_integer([4, big, 6, 2]),
                       chip.assign_integer([0, 5, n_ok, (1 << 2047) + 1]))
    torch.cuda.synchronize()
    st = res.status.cpu().tolist()
    assert st[0] == H.H2R_E_ZERO_MODULUS and st[1] == H.H2R_E_NOT_REDUCED and st[2] == 0
    # big*2 / (2^2047+1) = 3 with a remainder: fits -> ok
    assert st[3] == 0 and res.value.to_big_uint()[3] == (big * 2) % ((1 << 2047) + 1)
    rsa = H.RSAChip(2048, 5)
    pk = rsa.assign_public_key(H.RSAPublicKey(H.UnassignedInteger.from_ints([n_ok, n_ok, n_ok], 32, 64), H.Fix(65537)))
    x = rsa.bigint_chip().assign_integer([n_ok - 1, n_ok, n_ok + 1])
    res = rsa.modpow_public_key(x, pk, want_trace=False)
    torch.cuda.synchronize()
    assert res.status.cpu().tolist() == [0, H.H2R_E_NOT_IN_FIELD, H.H2R_E_NOT_IN_FIELD]
    assert res.value.to_big_uint()[0] == pow(n_ok - 1, 65537, n_ok)
    # the same on the RSAPubE::Var arm (the reference asserts x < n before BOTH arms, src/chip.rs:106), where the
    # in-field witness of a failing element is still produced (is_less_than = 0 in its stream)
    pkv = rsa.assign_public_key(H.RSAPublicKey(H.UnassignedInteger.from_ints([n_ok, n_ok, n_ok], 32, 64),
                                               H.Var(H.UnassignedInteger.from_ints([17, 17, 17], 1, 64))))
    res = rsa.modpow_public_key(x, pkv)
    torch.cuda.synchronize()
    assert res.status.cpu().tolist() == [0, H.H2R_E_NOT_IN_FIELD, H.H2R_E_NOT_IN_FIELD]
    assert res.value.to_big_uint()[0] == pow(n_ok - 1, 17, n_ok)
    o = Oracle(64, 32)
    for i, xv in enumerate([n_ok - 1, n_ok, n_ok + 1]):
        rc, lt, s_if = o.assert_in_field(o.limbs(xv), o.limbs(n_ok))
        assert lt == (1 if i == 0 else 0) and np.array_equal(res.in_field.flatten(i), s_if)
    # pow_mod itself (BigIntInstructions, no in-field assertion) accepts x >= n as long as the quotients fit
    res = chip.pow_mod(x, pkv.e.e, pkv.n, 5)
    torch.cuda.synchronize()
    assert res.status.cpu().tolist()[:2] == [0, 0] and res.value.to_big_uint()[1] == 0
    # an exponent limb with bits at or above exp_limb_bits cannot satisfy main_gate.to_bits (big_integer/chip.rs:677)
    pkw = rsa.assign_public_key(H.RSAPublicKey(H.UnassignedInteger.from_ints([n_ok, n_ok, n_ok], 32, 64),
                                               H.Var(H.UnassignedInteger.from_ints([31, 32, 1 << 63], 1, 64))))
    res = rsa.modpow_public_key(rsa.bigint_chip().assign_integer([5, 5, 5]), pkw, want_trace=False)
    torch.cuda.synchronize()
    assert res.status.cpu().tolist() == [0, H.H2R_E_SHAPE, H.H2R_E_SHAPE]
    assert o.pow_mod(o.limbs(5), np.array([32], np.uint64), 5, o.limbs(n_ok))[0] == 1   # the oracle refuses it too (H2RO_E_SHAPE)


@pytest.mark.parametrize("w,L,batch,e", [(64, 32, 12, 65537), (64, 16, 8, 65537), (32, 128, 3, 65537), (64, 32, 4, 0b1011011), (64, 32, 3, 1), (64, 64, 2, 17),
                                         (64, 48, 6, 65537), (64, 24, 8, 65537), (32, 96, 3, 65537), (64, 36, 4, 0b1011011)])
def test_pow_mod_fixed_exp_parity(H, w, L, batch, e):
    """pow_mod_fixed_exp (reference big_integer/chip.rs:710-742; tests :2314-2353 use a 7-bit e)."""
    chip = H.BigIntChip(w, w * L)
    o = Oracle(w, L)
    rng = random.Random(w * 1000 + L + e)
    N = [rand_modulus(rng, w * L, odd=(i % 3 != 2)) for i in range(batch)]
    X = [rng.randrange(n) for n in N]
    res = chip.pow_mod_fixed_exp(chip.assign_integer(X), e, chip.assign_integer(N))
    torch.cuda.synchronize()
    assert not res.status.cpu().numpy().any()
    out = res.value.to_big_uint()
    for i in range(batch):
        assert out[i] == pow(X[i], e, N[i])
        rc, oo, ost = o.pow_mod_fixed_exp(o.limbs(X[i]), o.limbs(N[i]), e)
        assert rc == 0 and np.array_equal(ost, res.trace.flatten(i)), (w, L, i)


def test_rsa4096_w32_golden(H, golden):
    """BASELINE config 4 shape (128 x 32-bit limbs): golden stream digest."""
    c = golden["rsa4096_w32"]
    chip = H.BigIntChip(32, 4096)
    res = chip.pow_mod_fixed_exp(chip.assign_integer([int(c["x"])]), c["e"], chip.assign_integer([int(c["n"])]))
    torch.cuda.synchronize()
    assert res.value.to_big_uint()[0] == int(c["result"])
    st = res.trace.flatten(0)
    assert len(st) == c["stream_bytes"] and sha(st) == c["stream_sha256"]


def test_pow_mod_var_parity(H, golden):
    """pow_mod with a 5-bit variable exponent (reference big_integer/chip.rs:664-696; src/chip.rs:283, 327)."""
    rsa = H.RSAChip(2048, 5)
    chip = rsa.bigint_chip()
    o = Oracle(64, 32)
    k = golden["rsa_kats"][0]
    es = [c["e"] for c in golden["pow_var_kat1"]] + [7, 24]
    rng = random.Random(99)
    N = [int(k["n"])] * 4 + [rand_modulus(rng, 2048), rand_modulus(rng, 2048, odd=False)]
    X = [int(k["sig"])] * 4 + [rng.randrange(N[4]), rng.randrange(N[5])]
    pk = rsa.assign_public_key(H.RSAPublicKey(H.UnassignedInteger.from_ints(N, 32, 64), H.Var(H.UnassignedInteger.from_ints(es, 1, 64))))
    res = rsa.modpow_public_key(chip.assign_integer(X), pk)
    torch.cuda.synchronize()
    assert not res.status.cpu().numpy().any()
    for i in range(len(es)):
        assert res.value.to_big_uint()[i] == pow(X[i], es[i], N[i])
        rc, oo, ost = o.pow_mod(o.limbs(X[i]), np.array([es[i]], np.uint64), 5, o.limbs(N[i]))
        st = res.trace.flatten(i)
        assert rc == 0 and np.array_equal(ost, st)
        if i < 4:
            assert sha(st) == golden["pow_var_kat1"][i]["stream_sha256"]


@pytest.mark.parametrize("w,L,e_num_limbs,exp_limb_bits,batch", [
    (64, 32, 2, 33, 3),     # bits of a limb straddle the 32-bit word fetch
    (64, 32, 2, 64, 2),     # whole 64-bit limbs, two of them
    (64, 16, 32, 3, 2),     # many narrow limbs
    (64, 16, 3, 17, 3),
    (32, 32, 2, 32, 2),     # 32-bit limbs: one digit per limb
    (32, 32, 5, 7, 2),
    (64, 8, 32, 64, 2),     # 2048 exponent bits on a 512-bit modulus (4,096 mul_mods per element)
])
def test_pow_mod_var_multi_limb(H, w, L, e_num_limbs, exp_limb_bits, batch):
    """pow_mod (reference big_integer/chip.rs:664-696) with several exponent limbs and wide exp_limb_bits: the
    per-limb main_gate.to_bits order (:674-681), every mul_mod / select stream and the result, byte-exact vs the oracle."""
    chip = H.BigIntChip(w, w * L)
    o = Oracle(w, L)
    rng = random.Random(1000 * w + 31 * L + 7 * e_num_limbs + exp_limb_bits)
    N = [rand_modulus(rng, w * L, odd=(i != 1)) for i in range(batch)]
    X = [rng.randrange(n) for n in N]
    E = [[rng.getrandbits(exp_limb_bits) for _ in range(e_num_limbs)] for _ in range(batch)]
    E[0][0] |= 1 << (exp_limb_bits - 1)      # top bit of a limb set
    if e_num_limbs > 1:
        E[0][1] = 0                              # an all-zero limb still costs exp_limb_bits iterations
    ei = H.UnassignedInteger(np.array(E, dtype=chip.np_dtype))
    e_dev = H.AssignedInteger(torch.from_numpy(ei.limbs.view(np.int64 if w == 64 else np.int32)).cuda().contiguous(), w)
    res = chip.pow_mod(chip.assign_integer(X), e_dev, chip.assign_integer(N), exp_limb_bits)
    torch.cuda.synchronize()
    assert not res.status.cpu().numpy().any()
    out = res.value.to_big_uint()
    for i in range(batch):
        e_int = sum(v << (exp_limb_bits * k) for k, v in enumerate(E[i]))
        assert out[i] == pow(X[i], e_int, N[i]), i
        rc, oo, ost = o.pow_mod(o.limbs(X[i]), np.array(E[i], dtype=o.dtype), exp_limb_bits, o.limbs(N[i]))
        assert rc == 0 and o.to_int(oo) == out[i]
        st = res.trace.flatten(i)
        if not np.array_equal(ost, st):
            pytest.fail("elem %d: first stream mismatch at byte %d of %d" % (i, int(np.nonzero(ost != st)[0][0]), len(st)))


def test_pow_mod_var_2048_bit_exponent(H):
    """SURVEY a3 "C5 alternative": RSA-2048 with a 2,048-bit VARIABLE exponent (32 limbs x 64 bits): 4,096 mul_mods
    per element (263 MB of trace each).  Result vs pow(); the whole flat stream of one element vs the oracle."""
    chip = H.BigIntChip(64, 2048)
    o = Oracle(64, 32)
    rng = random.Random(0x68327273 + 50)
    N = [rand_modulus(rng, 2048) for _ in range(2)]
    X = [rng.randrange(n) for n in N]
    E = [[rng.getrandbits(64) for _ in range(32)] for _ in range(2)]
    E[0][31] |= 1 << 63
    e_dev = H.AssignedInteger(torch.from_numpy(np.array(E, dtype=np.uint64).view(np.int64)).cuda().contiguous(), 64)
    res = chip.pow_mod(chip.assign_integer(X), e_dev, chip.assign_integer(N), 64)
    torch.cuda.synchronize()
    assert not res.status.cpu().numpy().any()
    out = res.value.to_big_uint()
    for i in range(2):
        assert out[i] == pow(X[i], sum(v << (64 * k) for k, v in enumerate(E[i])), N[i])
    rc, oo, ost = o.pow_mod(o.limbs(X[0]), np.array(E[0], dtype=np.uint64), 64, o.limbs(N[0]))
    assert rc == 0 and np.array_equal(ost, res.trace.flatten(0))


@pytest.mark.parametrize("w,L,batch", [(64, 32, 9), (32, 128, 3), (64, 64, 3), (64, 48, 4), (64, 12, 11), (64, 4, 20), (32, 8, 20), (32, 96, 3),
                                       (64, 16, 7)])
@pytest.mark.parametrize("flags", [0, 1])
def test_device_emit_stream_mul_mod(H, w, L, batch, flags):
    """h2r_trace_emit_stream: the flat stream of every record, produced on the device, is byte-equal to the host walk
    (h2r_trace_flatten_ex) and -- for flags = 0 -- to the oracle; with H2R_STREAM_FIELD_AB to the Python restatement run
    with the field modulus (a_b = p - |x| when negative, big_integer/chip.rs:859).  Unaligned output strides included."""
    import os
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
    import pyref as R
    field = ["bn254_fr", "pasta_fq"][(w + L) % 2]
    chip = H.BigIntChip(w, w * L, field=field)
    o = Oracle(w, L)
    rng = random.Random(5 * w + L + flags)
    N = [rand_modulus(rng, w * L, odd=(i % 4 != 1)) for i in range(batch)]
    A = [rng.randrange(n) for n in N]
    B = [rng.randrange(n) for n in N]
    res = chip.mul_mod(chip.assign_integer(A), chip.assign_integer(B), chip.assign_integer(N))
    sb = res.trace.stream_bytes_ex(flags)
    for stride in (sb, sb + 7, (sb + 255) // 256 * 256):   # odd strides: every record starts at another byte phase
        out = torch.full((batch, stride), 0xEE, dtype=torch.uint8, device="cuda")
        res.trace.emit_stream(flags, out=out, out_stride=stride)
        torch.cuda.synchronize()
        host = out.cpu().numpy()
        assert (host[:, sb:] == 0xEE).all()       # nothing written past an element's stream
        for i in range(batch):
            want = res.trace.flatten(i, flags)
            if not np.array_equal(host[i, :sb], want):
                pytest.fail("w=%d L=%d flags=%d stride=%d elem %d: first mismatch at byte %d of %d" %
                            (w, L, flags, stride, i, int(np.nonzero(host[i, :sb] != want)[0][0]), sb))
    for i in range(min(batch, 3)):
        st = R.Stream()
        R.mul_mod(R.Params(w, L, field_modulus=R.FIELD_MODULI[field] if flags else 0), R.to_limbs(A[i], L, w), R.to_limbs(B[i], L, w),
                  R.to_limbs(N[i], L, w), st)
        assert bytes(host[i, :sb]) == st.bytes()


@pytest.mark.parametrize("w,L,e,var_bits", [(64, 32, 65537, 0), (64, 16, 0b1011011, 0), (32, 128, 17, 0), (64, 32, 0, 5), (64, 8, 0, 13),
                                            (64, 24, 65537, 0), (64, 32, 1, 0)])
def test_device_emit_stream_pow(H, w, L, e, var_bits):
    """h2r_pow_trace_emit_stream == h2r_pow_trace_flatten_ex for every element: fixed exponents (records back to back,
    then the result limbs) and variable exponents (e bits, then per bit mul_mod record / selected limbs / square record)."""
    chip = H.BigIntChip(w, w * L)
    rng = random.Random(w + L + e + var_bits)
    batch = 5
    N = [rand_modulus(rng, w * L) for _ in range(batch)]
    X = [rng.randrange(n) for n in N]
    if var_bits:
        E = np.array([[rng.getrandbits(var_bits)] for _ in range(batch)], dtype=chip.np_dtype)
        e_dev = H.AssignedInteger(torch.from_numpy(E.view(np.int64 if w == 64 else np.int32)).cuda().contiguous(), w)
        res = chip.pow_mod(chip.assign_integer(X), e_dev, chip.assign_integer(N), var_bits)
    else:
        res = chip.pow_mod_fixed_exp(chip.assign_integer(X), e, chip.assign_integer(N))
    torch.cuda.synchronize()
    assert not res.status.cpu().numpy().any()
    for flags in (0, 1):
        sb = res.trace.stream_bytes_ex(flags)
        out = res.trace.emit_stream(flags, out_stride=sb + 3)
        torch.cuda.synchronize()
        host = out.cpu().numpy()
        for i in range(batch):
            want = res.trace.flatten(i, flags)
            if not np.array_equal(host[i, :sb], want):
                pytest.fail("elem %d flags %d: first mismatch at byte %d of %d" % (i, flags, int(np.nonzero(host[i, :sb] != want)[0][0]), sb))


@pytest.mark.parametrize("w,L,field", [(64, 32, "bn254_fr"), (32, 128, "pasta_fp"), (64, 12, "bn254_fq"), (64, 48, "pasta_fq"), (32, 8, "bn254_fr")])
def test_advice_image(H, w, L, field):
    """h2r_*_emit_advice: the COMPLETE 5-column advice image -- every cell the reference's ops assign: the flat stream's
    values, the assign_constant cells, the assign_bit(1) seeds, main_gate.is_zero's difference / inverse witnesses -- equals
    the image built in Python from the ORACLE's flat stream with the documented row table (tests/advice_ref.py), for a
    mul_mod batch and for a pow trace (with its two constant rows); and it is a SATISFYING assignment: every row fulfils the
    main-gate equation with the fixed row the C ABI reports for its kind (h2r_advice_row_kinds / h2r_advice_fixed_row)."""
    import ctypes
    import os
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
    import pyref as R
    import advice_ref as AR
    from halo2_rsa_amd import _lib
    from halo2_rsa_amd._lib import lib
    chip = H.BigIntChip(w, w * L, field=field)
    o = Oracle(w, L)
    P = R.FIELD_MODULI[field]
    rng = random.Random(3 * w + L)
    batch = 3
    N = [rand_modulus(rng, w * L, odd=(i != 1)) for i in range(batch)]
    A = [rng.randrange(n) for n in N]
    Bv = [rng.randrange(n) for n in N]
    res = chip.mul_mod(chip.assign_integer(A), chip.assign_integer(Bv), chip.assign_integer(N))
    img = res.emit_advice()
    torch.cuda.synchronize()
    rows = int(lib().h2r_advice_rows(chip._ctx))
    assert img.shape == (batch, rows * 160)
    if (w, L) == (64, 32):
        assert rows == 3974
    # the fixed side, from the C ABI: row kinds and the selectors of every kind
    kinds = np.zeros(rows, dtype=np.uint8)
    assert lib().h2r_advice_row_kinds(chip._ctx, kinds.ctypes.data) == 0
    la = H.LookupArgument(chip, rsa_chip=(w == 64))
    cfg = AR.LookupConfig(AR.range_lens(w, L, rsa=(w == 64)))
    fixed = {}
    for k in sorted(set(kinds.tolist())):
        fr = _lib.H2RFixedRow()
        assert lib().h2r_advice_fixed_row(chip._ctx, ctypes.byref(la.cfg), k, ctypes.byref(fr)) == 0
        fixed[k] = fr.as_dict()
        ref = AR.fixed_row(k, w, L, o.p.carry_bits, o.p.carry_sub_bits, o.p.carry_nsub, cfg)
        assert {nm: v % P for nm, v in ref.items() if nm in AR.FIXED_NAMES} == {nm: fixed[k][nm] for nm in AR.FIXED_NAMES}, k
        assert (ref["tag_composition"], ref["tag_overflow"]) == (fixed[k]["tag_composition"], fixed[k]["tag_overflow"])
    host = img.cpu().numpy()
    table = set(cfg.table())
    for i in range(batch):
        rc, rr, ost = o.mul_mod(o.limbs(A[i]), o.limbs(Bv[i]), o.limbs(N[i]))
        im = AR.mul_mod_image(o.p, [int(v) for v in o.limbs(A[i])], [int(v) for v in o.limbs(Bv[i])], [int(v) for v in o.limbs(N[i])], ost, P)
        assert im.kinds == kinds.tolist()
        want = AR.image_bytes(im)
        assert want.shape == (rows, 160)
        got = host[i].reshape(rows, 160)
        if not np.array_equal(got, want):
            bad = np.argwhere(got != want)[0]
            pytest.fail("w=%d L=%d elem %d: row %d (kind %d) cell %d differs" % (w, L, i, int(bad[0]), int(kinds[int(bad[0])]), int(bad[1]) // 32))
        if i == 0:   # gate equation + lookup membership on the GPU's own cells
            cells = [[int.from_bytes(got[r, 32 * c:32 * c + 32].tobytes(), "little") for c in range(5)] for r in range(rows)]
            for r in range(rows):
                f = fixed[int(kinds[r])]
                assert AR.gate_residual(cells[r], cells[r + 1][4] if r + 1 < rows else 0, f, P) == 0, (r, int(kinds[r]))
                if f["tag_composition"]:
                    assert all((f["tag_composition"], cells[r][c]) in table for c in range(4)), r
                if f["tag_overflow"]:
                    assert (f["tag_overflow"], cells[r][0]) in table, r
    # the records of a pow trace (operands from the call's workspace), behind the two constant rows of acc = 1
    e = 0b1011
    pres = chip.pow_mod_fixed_exp(chip.assign_integer(A), e, chip.assign_integer(N))
    pimg = pres.emit_advice().cpu().numpy()
    T = pres.trace.num_mul_mods
    assert pimg.shape[1] == (2 + T * rows) * 160
    one = np.zeros((2, 160), dtype=np.uint8)
    one[0, 0] = 1
    assert np.array_equal(pimg[0, :320].reshape(2, 160), one)
    rc, oo, ost = o.pow_mod_fixed_exp(o.limbs(A[0]), o.limbs(N[0]), e)
    msb = o.mul_mod_stream_bytes
    acc, cur, t = 1, A[0], 0
    for bit in [(e >> k) & 1 for k in range(e.bit_length())]:
        ops = [(cur, cur)] + ([(acc, cur)] if bit else [])
        nxt = cur * cur % N[0]
        for (x, y) in ops:
            want = AR.advice_image_from_stream(o.p, [int(v) for v in o.limbs(x)], [int(v) for v in o.limbs(y)],
                                               [int(v) for v in o.limbs(N[0])], ost[t * msb:(t + 1) * msb], P)
            got = pimg[0, (2 + t * rows) * 160:(2 + (t + 1) * rows) * 160].reshape(rows, 160)
            assert np.array_equal(got, want), ("pow record", t)
            t += 1
        if bit:
            acc = acc * cur % N[0]
        cur = nxt
    assert t == T


@pytest.mark.parametrize("w,L,field", [(64, 32, "bn254_fr"), (32, 8, "pasta_fq")])
def test_advice_image_is_zero_inverse_witness(H, w, L, field):
    """main_gate.is_equal on UNEQUAL values (never the case in a valid mul_mod): corrupt the stored mod_acc, the carry duplicate and
    the final acc_extra of a record and rebuild the image -- the is_equal rows must then hold d = x - y != 0 (a field element,
    p - |d| when negative), its inverse d^-1 and r from the (unchanged) flag bytes; the rows [d, 1/d, r] must satisfy
    d * (1/d) + r - 1 = 0 with r = 0, i.e. the inverse is right.  Exercises the kernel's field inversion (380 Montgomery products)."""
    import ctypes
    import os
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
    import pyref as R
    from halo2_rsa_amd._lib import lib
    chip = H.BigIntChip(w, w * L, field=field)
    P = R.FIELD_MODULI[field]
    rng = random.Random(w + L)
    n = rand_modulus(rng, w * L)
    a, b = rng.randrange(n), rng.randrange(n)
    res = chip.mul_mod(chip.assign_integer([a]), chip.assign_integer([b]), chip.assign_integer([n]))
    torch.cuda.synchronize()
    lo = chip.layout
    from halo2_rsa_amd import _lib
    P_IDX = {nm: k for k, nm in enumerate(_lib.PLANES)}
    buf = res.trace.buf
    LB, CB = lo.limb_bytes, lo.carry_bytes
    C = 2 * L - 1

    def bump(plane, idx, nbytes, delta):   # little-endian += delta on a stored value
        off = lo.plane_off[P_IDX[plane]] + idx * lo.plane_elem[P_IDX[plane]]
        v = int.from_bytes(buf[off:off + nbytes].cpu().numpy().tobytes(), "little") + delta
        buf[off:off + nbytes] = torch.from_numpy(np.frombuffer((v % (1 << (8 * nbytes))).to_bytes(nbytes, "little"), dtype=np.uint8).copy()).to(buf.device)
        return v

    vals = {}
    for nm, idx, nb in (("CMOD", 1, LB), ("MODACC", 1, LB), ("CARRY", 2, CB), ("CARRY_DUP", 2, CB), ("CARRY", C - 1, CB), ("QACC", C - 1, CB)):
        off = lo.plane_off[P_IDX[nm]] + idx * lo.plane_elem[P_IDX[nm]]
        vals[(nm, idx)] = int.from_bytes(buf[off:off + nb].cpu().numpy().tobytes(), "little")
    bump("MODACC", 1, LB, 5)                 # c - mod_acc = -5 (or +2^w - 5 after wrap: use the stored values below)
    bump("CARRY_DUP", 2, CB, -3)             # carry - dup = +3
    bump("QACC", C - 1, CB, 1)               # final carry - acc_extra = -1
    img = res.emit_advice().cpu().numpy()
    rows = int(lib().h2r_advice_rows(chip._ctx))
    got = img[0].reshape(rows, 160)
    cell = lambda r, c: int.from_bytes(got[r, 32 * c:32 * c + 32].tobytes(), "little")
    nrc = (lo.carry_nsub + 3) // 4
    r_T6 = 4 * L + 2 * (C + L * L) + L + 4
    per_col = 23 + nrc

    def check(col, k_sub, x, y):
        r0 = r_T6 + col * per_col + k_sub + (nrc if (k_sub == 18 and col < C - 1) else 0)
        d = (x - y) % P
        assert d != 0
        assert [cell(r0, 0), cell(r0, 1), cell(r0, 2)] == [x, y, d], (col, k_sub)
        assert cell(r0 + 2, 0) == d and (cell(r0 + 2, 0) * cell(r0 + 2, 1)) % P == 1, (col, k_sub)       # d * d^-1 = 1
        assert cell(r0 + 2, 1) == pow(d, P - 2, P)
        assert cell(r0 + 3, 1) == d
    mod1 = (vals[("MODACC", 1)] + 5) % (1 << (8 * LB))
    check(1, 13, vals[("CMOD", 1)], mod1)
    dup2 = (vals[("CARRY_DUP", 2)] - 3) % (1 << (8 * CB))
    check(2, 18, vals[("CARRY", 2)], dup2)
    qlast = vals[("QACC", C - 1)] + 1
    check(C - 1, 18, vals[("CARRY", C - 1)], qlast)


def test_shared_modulus(H):
    """One key, many signatures (H2R_F_SHARED_MODULUS)."""
    chip = H.BigIntChip(64, 2048)
    rng = random.Random(3)
    n = rand_modulus(rng, 2048)
    X = [rng.randrange(n) for _ in range(5)]
    res = chip.pow_mod_fixed_exp(chip.assign_integer(X), 65537, chip.assign_integer([n]))
    res2 = chip.pow_mod_fixed_exp(chip.assign_integer(X), 65537, chip.assign_integer([n] * 5))
    torch.cuda.synchronize()
    assert res.value.to_big_uint() == [pow(x, 65537, n) for x in X]
    assert torch.equal(res.trace.buf, res2.trace.buf) or all(np.array_equal(res.trace.flatten(i), res2.trace.flatten(i)) for i in range(5))


def expected_hist(o, streams):
    """Count the sub-limb lookups of oracle mul_mod streams: rows = 2^limb_sub_bits composition,
    [2^carry_sub_bits composition if different], 2^overflow_bits overflow."""
    p = o.p
    L, C = p.L, 2 * p.L - 1
    t0 = 1 << p.limb_sub_bits
    same = p.carry_sub_bits == p.limb_sub_bits
    t1_off = 0 if same else t0
    t1_len = 0 if same else (1 << p.carry_sub_bits)
    ovb = p.carry_bits % p.carry_sub_bits
    t2_off = t0 + t1_len
    hist = np.zeros(t2_off + ((1 << ovb) if ovb else 0), dtype=np.int64)
    per_col = 5 * p.WB + 2 * p.CB + 4 * p.LB + 2
    for st in streams:
        pos = 0
        for _ in range(2 * L):
            pos += p.LB
            for _ in range(p.limb_nsub):
                hist[int(st[pos])] += 1
                pos += 1
        pos += 2 * L * L * p.WB + L * p.WB
        for i in range(C):
            pos += per_col
            if i < C - 1:
                pos += p.CB
                for j in range(p.carry_nsub):
                    if ovb and j == p.carry_nsub - 1:
                        hist[t2_off + int(st[pos])] += 1
                    else:
                        hist[t1_off + int(st[pos])] += 1
                    pos += 1
            pos += 2
        assert pos == len(st)
    return hist


@pytest.mark.parametrize("w,L", [(64, 32), (32, 128)])
def test_lookup_multiplicities(H, w, L):
    """The range-check lookup batch: per-circuit multiplicity of every (tag, value) table row
    (RangeChip call sites big_integer/chip.rs:590, 598, 880-885)."""
    chip = H.BigIntChip(w, w * L)
    o = Oracle(w, L)
    rng = random.Random(11)
    N = [rand_modulus(rng, w * L) for _ in range(3)]
    X = [rng.randrange(n) for n in N]
    res = chip.pow_mod_fixed_exp(chip.assign_integer(X), 65537, chip.assign_integer(N))
    hist = res.trace.lookup_hist().cpu().numpy()
    msb = o.mul_mod_stream_bytes
    for i in range(3):
        rc, oo, ost = o.pow_mod_fixed_exp(o.limbs(X[i]), o.limbs(N[i]), 65537)
        streams = [ost[t * msb:(t + 1) * msb] for t in range(19)]
        want = expected_hist(o, streams)
        assert np.array_equal(hist[i], want)
        if (w, L) == (64, 32):
            assert int(hist[i].sum()) == 20330   # sub-limb lookups per RSA-2048 e=65537 pow path


def test_range_decompose_batch(H):
    import ctypes
    from halo2_rsa_amd._lib import check, lib
    chip = H.BigIntChip(64, 2048)
    rng = np.random.default_rng(5)
    vals = rng.integers(0, 1 << 63, size=(1000, 2), dtype=np.uint64)
    vals[:, 1] &= np.uint64(0x3f)            # 70-bit values
    dv = torch.from_numpy(vals.view(np.int64)).cuda()
    sub = torch.zeros((1000, 12), dtype=torch.uint8, device="cuda")
    hist = torch.zeros(256 + 64, dtype=torch.int32, device="cuda")
    check(lib().h2r_range_decompose_batch(chip._ctx, dv.data_ptr(), 16, 1000, 70, 8, sub.data_ptr(), 12, hist.data_ptr(), None), "decompose")
    torch.cuda.synchronize()
    s = sub.cpu().numpy()
    want = np.zeros((1000, 9), dtype=np.uint8)
    for i in range(1000):
        v = int(vals[i, 0]) | (int(vals[i, 1]) << 64)
        want[i] = [(v >> (8 * k)) & 0xff for k in range(9)]
    assert np.array_equal(s[:, :9], want)
    h = hist.cpu().numpy()
    assert np.array_equal(h[:256], np.bincount(want[:, :8].ravel(), minlength=256))
    assert np.array_equal(h[256:], np.bincount(want[:, 8], minlength=64))


def test_config2_full_batch_properties(H, golden):
    """BASELINE config 2 at full size (batch 1024, RSA-2048, e = 65537, KAT1/KAT2/BAD as elements 0-2):
    every result equals pow(x, e, n); ALL 1,024 elements are byte-exact vs the oracle; sampled records'
    q/r planes satisfy a*b = q*n + r."""
    chip = H.BigIntChip(64, 2048)
    o = Oracle(64, 32)
    rng = random.Random(0x68327273 + 2)
    kats = golden["rsa_kats"]
    N = [int(k["n"]) for k in kats] + [rand_modulus(rng, 2048) for _ in range(1021)]
    X = [int(k["sig"]) for k in kats] + [rng.randrange(n) for n in N[3:]]
    res = chip.pow_mod_fixed_exp(chip.assign_integer(X), 65537, chip.assign_integer(N))
    torch.cuda.synchronize()
    assert not res.status.cpu().numpy().any()
    out = res.value.to_big_uint()
    assert all(out[i] == pow(X[i], 65537, N[i]) for i in range(1024))
    # EVERY element's flat stream, byte for byte, against the threaded C oracle (128 elements at a time: one D2H copy
    # of their records, h2r_pow_trace_flatten on the host copy)
    import os
    from halo2_rsa_amd._lib import check, lib
    xs, nsl = chip.assign_integer(X).limbs_host(), chip.assign_integer(N).limbs_host()
    pl, es = res.trace.pow_layout, res.trace.elem_stride
    got = np.zeros(pl.stream_bytes, dtype=np.uint8)
    dev_stream = res.trace.emit_stream()      # the device-side flatten of the whole batch (h2r_pow_trace_emit_stream)
    torch.cuda.synchronize()
    assert dev_stream.shape == (1024, pl.stream_bytes)
    for lo in range(0, 1024, 128):
        oout, ostat, ost = o.pow_mod_fixed_exp_batch(xs[lo:lo + 128], nsl[lo:lo + 128], 65537, nthreads=min(64, os.cpu_count() or 1),
                                                     want_stream=True)
        assert not ostat.any()
        host = res.trace.buf[lo * es:(lo + 128) * es].cpu().numpy()
        for k in range(128):
            check(lib().h2r_pow_trace_flatten(chip._ctx, ctypes.byref(pl), host[k * es:(k + 1) * es].ctypes.data, got.ctypes.data), "flatten")
            if not np.array_equal(got, ost[k]):
                pytest.fail("element %d: first stream mismatch at byte %d" % (lo + k, int(np.nonzero(got != ost[k])[0][0])))
        demit = dev_stream[lo:lo + 128].cpu().numpy()
        if not np.array_equal(demit, ost):
            bad = np.argwhere(demit != ost)[0]
            pytest.fail("device-side stream: element %d differs at byte %d" % (lo + int(bad[0]), int(bad[1])))
    # chain identity on the q/r planes of every record of 64 more elements
    for i in rng.sample(range(1024), 64):
        acc, cur, t = 1, X[i], 0

        def qr(t):
            q = int.from_bytes(res.trace.plane(i, t, "Q").tobytes(), "little")
            r = int.from_bytes(res.trace.plane(i, t, "R").tobytes(), "little")
            return q, r
        for k in range(17):
            q, r = qr(t)
            assert cur * cur == q * N[i] + r and r < N[i]
            nxt = r
            t += 1
            if (65537 >> k) & 1:
                q, r = qr(t)
                assert acc * cur == q * N[i] + r and r < N[i]
                acc = r
                t += 1
            cur = nxt
        assert t == 19 and acc == out[i]


def test_verify_pkcs1v15_signature_kats(H, golden):
    """RSAChip::verify_pkcs1v15_signature after the SHA step (reference src/chip.rs:683-816):
    is_valid = 1, 1, 0 for KAT1, KAT2, BAD; the whole witness (assert_in_field + pow + encoded-message
    check) byte-exact vs the oracle, and vs the golden digests minted from the reference's vectors."""
    rsa = H.RSAChip(2048, 5)
    kats = golden["rsa_kats"]
    rng = random.Random(21)
    ns = [int(k["n"]) for k in kats] + [rand_modulus(rng, 2048) for _ in range(5)]
    sigs = [int(k["sig"]) for k in kats] + [rng.randrange(n) for n in ns[3:]]
    hashed = [int(k["hashed"]) for k in kats] + [rng.getrandbits(256) for _ in range(5)]
    # element 7: a forged "signature" that decrypts to a well-formed EM for its hash (textbook RSA with a tiny key is
    # not available, so instead craft x = EM^(1/e) impossible) -> keep random; element 6: x >= n (not in field)
    sigs[6] = ns[6] + 5
    pk = rsa.assign_public_key(H.RSAPublicKey(H.UnassignedInteger.from_ints(ns, 32, 64), H.Fix(65537)))
    sg = rsa.assign_signature(H.RSASignature(H.UnassignedInteger.from_ints(sigs, 32, 64)))
    res = rsa.verify_pkcs1v15_signature(pk, hashed, sg)
    torch.cuda.synchronize()
    st = res.status.cpu().tolist()
    assert st[:6] == [0] * 6 and st[6] == H.H2R_E_NOT_IN_FIELD and st[7] == 0
    assert res.is_valid.cpu().tolist() == [1, 1, 0, 0, 0, 0, 0, 0]
    o = Oracle(64, 32)
    for i in range(8):
        rc_if, lt, s_if = o.assert_in_field(o.limbs(sigs[i]), o.limbs(ns[i]))
        assert lt == (1 if sigs[i] < ns[i] else 0)
        got = res.flatten(i)
        assert np.array_equal(got[:len(s_if)], s_if), ("in_field", i)
        if i == 6:
            continue
        rc, out, s_pow = o.pow_mod_fixed_exp(o.limbs(sigs[i]), o.limbs(ns[i]), 65537)
        rc, ok, s_em = o.pkcs1v15_em_check(out, o.limbs(hashed[i], 4))
        assert ok == int(res.is_valid[i].item())
        assert np.array_equal(got, np.concatenate([s_if, s_pow, s_em])), i
        if i < 3:
            k = kats[i]
            assert sha(s_if) == k["in_field_stream_sha256"] and sha(s_em) == k["em_stream_sha256"]
            assert sha(got[len(s_if):len(s_if) + len(s_pow)]) == k["pow_stream_sha256"]


def test_verify_large_plain_call_matches_small_calls(H, golden):
    """h2r_verify_pkcs1v15_batch with 2,304 signatures is walked as overlapping sub-batches inside the call; verdicts,
    statuses, results and sampled witness streams must equal what two 1,152-signature calls (one launch each) give."""
    rsa = H.RSAChip(2048, 5)
    kats = golden["rsa_kats"]
    rng = random.Random(29)
    B = 2304
    base = [rand_modulus(rng, 2048) for _ in range(24)]
    ns = [int(q["n"]) for q in kats] + [base[i % 24] ^ ((i // 24) << 90) | 1 for i in range(B - 3)]
    sigs = [int(q["sig"]) for q in kats] + [((base[i % 24] >> 7) * (i + 5)) % ns[i + 3] for i in range(B - 3)]
    hashed = [int(q["hashed"]) for q in kats] + [rng.getrandbits(256) for _ in range(B - 3)]
    sigs[1700] = ns[1700] + 3      # not in field, second sub-batch

    def run(lo, hi):
        pk = rsa.assign_public_key(H.RSAPublicKey(H.UnassignedInteger.from_ints(ns[lo:hi], 32, 64), H.Fix(65537)))
        sg = rsa.assign_signature(H.RSASignature(H.UnassignedInteger.from_ints(sigs[lo:hi], 32, 64)))
        return rsa.verify_pkcs1v15_signature(pk, hashed[lo:hi], sg)

    big = run(0, B)
    halves = [run(0, B // 2), run(B // 2, B)]
    torch.cuda.synchronize()
    assert big.is_valid.cpu().tolist()[:3] == [1, 1, 0]
    assert int(big.status[1700]) == H.H2R_E_NOT_IN_FIELD and int((big.status != 0).sum()) == 1
    for h, half in enumerate(halves):
        sl = slice(h * B // 2, (h + 1) * B // 2)
        assert torch.equal(big.is_valid[sl], half.is_valid) and torch.equal(big.status[sl], half.status)
        ok = (half.status == 0)
        assert torch.equal(big.powed.limbs_dev[sl][ok], half.powed.limbs_dev[ok])
        for i in (0, 1, 2, 511, 1023 - h * 1152 if h == 0 else 0, B // 2 - 1):
            if i < 0 or int(half.status[i]) != 0:
                continue
            assert np.array_equal(big.flatten(h * B // 2 + i), half.flatten(i)), (h, i)
    for i in (1023, 1024, 1151, 1152, B - 1):   # both sides of the first sub-batch boundary and of the half boundary
        h, j = divmod(i, B // 2)
        assert np.array_equal(big.flatten(i), halves[h].flatten(j)), i


def test_pipelined_verify_matches_batch_call(H, golden):
    """h2r_pipeline_verify_pkcs1v15: three pipelined verifier batches over two buffer sets produce, element for
    element, the bytes and verdicts of h2r_verify_pkcs1v15_batch (itself checked against the oracle above) and the
    oracle's stream for sampled elements."""
    rsa = H.RSAChip(2048, 5)
    chip = rsa.bigint_chip()
    o = Oracle(64, 32)
    kats = golden["rsa_kats"]
    rng = random.Random(23)
    B = 48
    pipe = H.Pipeline(chip, depth=2, side_streams=1)
    sets, calls, snaps = [], [], []
    for k in range(3):
        ns = [int(q["n"]) for q in kats] + [rand_modulus(rng, 2048) for _ in range(B - 3)]
        sigs = [int(q["sig"]) for q in kats] + [rng.randrange(n) for n in ns[3:]]
        hashed = [int(q["hashed"]) for q in kats] + [rng.getrandbits(256) for _ in range(B - 3)]
        if k == 1:
            sigs[7] = ns[7] + 1
        pk = rsa.assign_public_key(H.RSAPublicKey(H.UnassignedInteger.from_ints(ns, 32, 64), H.Fix(65537)))
        sg = rsa.assign_signature(H.RSASignature(H.UnassignedInteger.from_ints(sigs, 32, 64)))
        ref = rsa.verify_pkcs1v15_signature(pk, hashed, sg)
        hd = torch.from_numpy(H.UnassignedInteger.from_ints(hashed, 4, 64).limbs.view(np.int64)).cuda()
        calls.append(dict(ns=ns, sigs=sigs, hashed=hashed, ref=ref, n=chip.assign_integer(pk.n), s=chip.assign_integer(sg.c), h=hd))
    vl = calls[0]["ref"].layout
    for _ in range(2):
        sets.append(dict(trace=torch.empty(B * vl.elem_stride, dtype=torch.uint8, device="cuda"),
                         ws=torch.empty(chip.workspace_bytes(B, vl.pow.num_mul_mods), dtype=torch.uint8, device="cuda"),
                         powed=torch.empty((B, 32), dtype=torch.int64, device="cuda"),
                         valid=torch.zeros(B, dtype=torch.uint8, device="cuda"),
                         status=torch.zeros(B, dtype=torch.uint8, device="cuda")))
    for k, c in enumerate(calls):
        s = sets[k % 2]
        if k >= 2:
            snaps.append(tuple(t.clone() for t in (s["trace"], s["powed"], s["valid"], s["status"])))
        pipe.verify_pkcs1v15(c["s"], 65537, c["n"], c["h"], s["trace"], s["ws"], s["powed"], s["valid"], s["status"])
    pipe.join()
    for k in (1, 2):
        s = sets[k % 2]
        snaps.append(tuple(t.clone() for t in (s["trace"], s["powed"], s["valid"], s["status"])))
    torch.cuda.synchronize()
    for k, c in enumerate(calls):
        trace, powed, valid, status = snaps[k]
        ref = c["ref"]
        assert torch.equal(valid, ref.is_valid) and torch.equal(status, ref.status)
        assert valid.cpu().tolist()[:3] == [1, 1, 0]
        ok_rows = (status == 0).nonzero().flatten()
        assert torch.equal(powed[ok_rows], ref.powed.limbs_dev[ok_rows])
        es = vl.elem_stride
        got = H.rsa.VerifyResult(valid, H.AssignedInteger(powed, 64), status, trace, vl, chip)
        for i in (0, 1, 2, 7, B - 1):
            if int(status[i]) != 0:
                continue
            assert np.array_equal(got.flatten(i), ref.flatten(i)), (k, i)
        i = 5
        rc_if, lt, s_if = o.assert_in_field(o.limbs(c["sigs"][i]), o.limbs(c["ns"][i]))
        rc, out, s_pow = o.pow_mod_fixed_exp(o.limbs(c["sigs"][i]), o.limbs(c["ns"][i]), 65537)
        rc, ok, s_em = o.pkcs1v15_em_check(out, o.limbs(c["hashed"][i], 4))
        assert np.array_equal(got.flatten(i), np.concatenate([s_if, s_pow, s_em])), k
    pipe.close()


def _probable_prime(rng, bits):
    small = [3, 5, 7, 11, 13, 17, 19, 23, 29, 31, 37, 41, 43, 47, 53, 59, 61, 67, 71, 73, 79, 83, 89, 97]
    while True:
        c = rng.getrandbits(bits) | (1 << (bits - 1)) | 1
        if all(c % q for q in small) and all(pow(b, c - 1, c) == 1 for b in (2, 3, 5, 7)) and (c - 1) % 65537:
            return c


def test_pipelined_verify_rsa1024_two_queue(H):
    """[r6] The reference's bench shape (benches/bench.rs:369-407: RSA-1024 verification) through h2r_pipeline_verify_pkcs1v15 at 1,280 signatures per
    call on a pipeline with two side streams and three buffer sets -- the two-queue form with one-wave chains when the streams sit on three hardware
    queues.  Four rotating calls equal h2r_verify_pkcs1v15_batch element for element (verdicts, statuses, results, flat streams of sampled elements);
    elements 0 and 1 carry VALID signatures (a prime modulus p stands in for n: sig = EM^(e^-1 mod p-1) mod p), element 2 the same with a wrong hash."""
    rsa = H.RSAChip(1024, 5)
    chip = rsa.bigint_chip()
    rng = random.Random(1024)
    B, CALLS, depth = 1280, 4, 3
    pipe = H.Pipeline(chip, depth=depth, side_streams=2)
    prime = _probable_prime(rng, 1024)
    d = pow(65537, -1, prime - 1)
    head = int.from_bytes(b"\x00\x01" + b"\xff" * (128 - 3 - 51) + b"\x00" + bytes.fromhex("3031300d060960864801650304020105000420"), "big")
    base_n = [rand_modulus(rng, 1024) for _ in range(64)]
    calls = []
    for k in range(CALLS):
        ns = [prime, prime, prime] + [base_n[(i + k) % 64] for i in range(B - 3)]
        hashed = [rng.getrandbits(256) for _ in range(B)]
        sigs = [pow((head << 256) | hashed[i], d, prime) for i in range(3)] + [rng.randrange(n) for n in ns[3:]]
        hashed[2] ^= 1 << 77                                  # a signature over another digest
        if k == 2:
            sigs[9] = ns[9] + 3                               # not in field
        pk = rsa.assign_public_key(H.RSAPublicKey(H.UnassignedInteger.from_ints(ns, 16, 64), H.Fix(65537)))
        sg = rsa.assign_signature(H.RSASignature(H.UnassignedInteger.from_ints(sigs, 16, 64)))
        ref = rsa.verify_pkcs1v15_signature(pk, hashed, sg)
        hd = torch.from_numpy(H.UnassignedInteger.from_ints(hashed, 4, 64).limbs.view(np.int64)).cuda()
        calls.append(dict(ref=ref, n=chip.assign_integer(pk.n), s=chip.assign_integer(sg.c), h=hd, ns=ns, sigs=sigs, hashed=hashed))
    vl = calls[0]["ref"].layout
    sets = [dict(trace=torch.zeros(B * vl.elem_stride, dtype=torch.uint8, device="cuda"),
                 ws=torch.empty(chip.workspace_bytes(B, vl.pow.num_mul_mods), dtype=torch.uint8, device="cuda"),
                 powed=torch.zeros((B, 16), dtype=torch.int64, device="cuda"), valid=torch.zeros(B, dtype=torch.uint8, device="cuda"),
                 status=torch.zeros(B, dtype=torch.uint8, device="cuda")) for _ in range(depth)]
    snaps = {}
    for k, c in enumerate(calls):
        s = sets[k % depth]
        if k >= depth:
            snaps[k - depth] = tuple(t.clone() for t in (s["trace"], s["powed"], s["valid"], s["status"]))
        pipe.verify_pkcs1v15(c["s"], 65537, c["n"], c["h"], s["trace"], s["ws"], s["powed"], s["valid"], s["status"])
    pipe.join()
    for k in range(CALLS - depth, CALLS):
        s = sets[k % depth]
        snaps[k] = tuple(t.clone() for t in (s["trace"], s["powed"], s["valid"], s["status"]))
    torch.cuda.synchronize()
    for k, c in enumerate(calls):
        trace, powed, valid, status = snaps[k]
        ref = c["ref"]
        assert torch.equal(valid, ref.is_valid) and torch.equal(status, ref.status), k
        assert valid.cpu().tolist()[:3] == [1, 1, 0], k
        st = status.cpu().tolist()
        assert all(v == (H.H2R_E_NOT_IN_FIELD if (k == 2 and i == 9) else 0) for i, v in enumerate(st)), k
        ok_rows = (status == 0).nonzero().flatten()
        assert torch.equal(powed[ok_rows], ref.powed.limbs_dev[ok_rows]), k
        got = H.rsa.VerifyResult(valid, H.AssignedInteger(powed, 64), status, trace, vl, chip)
        for i in (0, 1, 2, 8, 640, B - 1):
            assert np.array_equal(got.flatten(i), ref.flatten(i)), (k, i)
        if k == 3:   # ... and the ORACLE's streams (assert_in_field + pow_mod_fixed_exp + the encoded-message check), a valid, a wrong-hash and a random element
            o = Oracle(64, 16)
            for i in (0, 2, 8):
                rc_if, lt, s_if = o.assert_in_field(o.limbs(c["sigs"][i]), o.limbs(c["ns"][i]))
                rc, out, s_pow = o.pow_mod_fixed_exp(o.limbs(c["sigs"][i]), o.limbs(c["ns"][i]), 65537)
                rc, ok, s_em = o.pkcs1v15_em_check(out, o.limbs(c["hashed"][i], 4))
                assert ok == (1 if i == 0 else 0)
                assert np.array_equal(got.flatten(i), np.concatenate([s_if, s_pow, s_em])), i
    pipe.close()


def test_pipelined_verify_folded_into_the_step_launch(H, golden):
    """h2r_pipeline_verify_pkcs1v15 at 1,024 signatures per call (one-launch steps): from the second call on the chain role of the
    step launch writes the verifier's in-field + encoded-message witness itself (step_kernel<..., FOLD>), the first call of the
    train keeps the kernel behind its chain kernel.  Four calls over two buffer sets, different signatures / digests per call,
    one element not in the field: verdicts, statuses and results equal h2r_verify_pkcs1v15_batch for EVERY element, element bytes
    for sampled ones (KATs, the rejected one, both ends), and every record passes the in-place audit."""
    rsa = H.RSAChip(2048, 5)
    chip = rsa.bigint_chip()
    kats = golden["rsa_kats"]
    rng = random.Random(29)
    B = 1024
    pipe = H.Pipeline(chip, depth=2, side_streams=1)
    calls = []
    for k in range(4):
        ns = [int(kats[i % 3]["n"]) for i in range(B)]
        sigs = [int(kats[i % 3]["sig"]) if i < 6 else rng.randrange(ns[i]) for i in range(B)]
        hashed = [int(kats[i % 3]["hashed"]) if (i + k) % 2 == 0 else rng.getrandbits(256) for i in range(B)]
        sigs[100 + k] = ns[100 + k] + 1                         # not in the field
        pk = rsa.assign_public_key(H.RSAPublicKey(H.UnassignedInteger.from_ints(ns, 32, 64), H.Fix(65537)))
        sg = rsa.assign_signature(H.RSASignature(H.UnassignedInteger.from_ints(sigs, 32, 64)))
        ref = rsa.verify_pkcs1v15_signature(pk, hashed, sg)
        hd = torch.from_numpy(H.UnassignedInteger.from_ints(hashed, 4, 64).limbs.view(np.int64)).cuda()
        calls.append(dict(ref=ref, n=chip.assign_integer(pk.n), s=chip.assign_integer(sg.c), h=hd, k=k))
    vl = calls[0]["ref"].layout
    sets = [dict(trace=torch.zeros(B * vl.elem_stride, dtype=torch.uint8, device="cuda"),
                 ws=torch.zeros(chip.workspace_bytes(B, vl.pow.num_mul_mods), dtype=torch.uint8, device="cuda"),
                 powed=torch.zeros((B, 32), dtype=torch.int64, device="cuda"), valid=torch.zeros(B, dtype=torch.uint8, device="cuda"),
                 status=torch.zeros(B, dtype=torch.uint8, device="cuda")) for _ in range(2)]
    for k, c in enumerate(calls):
        s = sets[k % 2]
        pipe.verify_pkcs1v15(c["s"], 65537, c["n"], c["h"], s["trace"], s["ws"], s["powed"], s["valid"], s["status"])
    pipe.join()
    torch.cuda.synchronize()
    for k in (2, 3):                                            # the last users of the two buffer sets: both were folded launches
        s, c = sets[k % 2], calls[k]
        ref = c["ref"]
        assert torch.equal(s["valid"], ref.is_valid) and torch.equal(s["status"], ref.status)
        st = s["status"].cpu().tolist()
        assert st[100 + k] == H.H2R_E_NOT_IN_FIELD and sum(1 for v in st if v) == 1
        v = s["valid"].cpu().tolist()
        assert v[:6] == [1 if ((i + k) % 2 == 0 and i % 3 != 2) else 0 for i in range(6)]
        ok_rows = (s["status"] == 0).nonzero().flatten()
        assert torch.equal(s["powed"][ok_rows], ref.powed.limbs_dev[ok_rows])
        got = H.rsa.VerifyResult(s["valid"], H.AssignedInteger(s["powed"], 64), s["status"], s["trace"], vl, chip)
        for i in (0, 1, 2, 3, 99 + k, 101 + k, 511, 512, B - 1):
            assert np.array_equal(got.flatten(i), ref.flatten(i)), (k, i)
        # the in-field region of the rejected element is written as well (the witness of the failed comparison)
        i = 100 + k
        es = vl.elem_stride
        a = s["trace"][i * es + vl.off_in_field:i * es + vl.off_in_field + vl.in_field_stream_bytes]
        b = ref.trace[i * es + vl.off_in_field:i * es + vl.off_in_field + vl.in_field_stream_bytes]
        assert torch.equal(a, b)
    pipe.close()


def test_verify_pkcs1v15_1024(H):
    """RSA-1024 (the reference bench's key size, benches/bench.rs:393-407): a genuinely valid signature built
    with a known factorisation, plus tampered variants."""
    p = (1 << 511) + 111          # not prime-checked: we only need d with e*d = 1 mod lcm-like exponent; use a real RSA construction below
    # Build a valid pkcs1v15 EM directly and "sign" with a private exponent of a small-prime-product modulus.
    import math
    rng = random.Random(4)

    def is_probable_prime(n):
        if n % 2 == 0:
            return False
        for a in (2, 3, 5, 7, 11, 13, 17, 19, 23, 29):
            if pow(a, n - 1, n) != 1:
                return False
        return True

    def gen_prime(bits):
        while True:
            c = rng.getrandbits(bits) | (1 << (bits - 1)) | (1 << (bits - 2)) | 1
            if is_probable_prime(c) and math.gcd(c - 1, 65537) == 1:
                return c
    pr, qr = gen_prime(512), gen_prime(512)
    n = pr * qr
    assert n.bit_length() == 1024
    d = pow(65537, -1, (pr - 1) * (qr - 1))
    digest = rng.getrandbits(256)
    em = (0x0001 << (1024 - 16)) | (((1 << (8 * (128 - 3 - 19 - 32))) - 1) << (8 * (1 + 19 + 32))) | \
         (int.from_bytes(bytes.fromhex("3031300d060960864801650304020105000420"), "big") << 256) | digest
    sig = pow(em, d, n)
    assert pow(sig, 65537, n) == em
    rsa = H.RSAChip(1024, 5)
    sigs = [sig, sig ^ 1, sig]
    hashes = [digest, digest, digest ^ (1 << 200)]
    pk = rsa.assign_public_key(H.RSAPublicKey(H.UnassignedInteger.from_ints([n] * 3, 16, 64), H.Fix(65537)))
    sg = rsa.assign_signature(H.RSASignature(H.UnassignedInteger.from_ints(sigs, 16, 64)))
    res = rsa.verify_pkcs1v15_signature(pk, hashes, sg)
    torch.cuda.synchronize()
    assert res.status.cpu().tolist() == [0, 0, 0]
    assert res.is_valid.cpu().tolist() == [1, 0, 0]
    o = Oracle(64, 16)
    for i in range(3):
        rc_if, lt, s_if = o.assert_in_field(o.limbs(sigs[i]), o.limbs(n))
        rc, out, s_pow = o.pow_mod_fixed_exp(o.limbs(sigs[i]), o.limbs(n), 65537)
        rc, ok, s_em = o.pkcs1v15_em_check(out, o.limbs(hashes[i], 4))
        assert ok == [1, 0, 0][i]
        assert np.array_equal(res.flatten(i), np.concatenate([s_if, s_pow, s_em])), i


def _oracle_lookup_keys(o, ost, T, hist_tab1_off, hist_tab2_off):
    """Table rows of every range-check sub-limb of a pow stream, in stream order (q, r sub-limbs then carries)."""
    p = o.p
    L, C = p.L, 2 * p.L - 1
    msb = o.mul_mod_stream_bytes
    per_col = 5 * p.WB + 2 * p.CB + 4 * p.LB + 2
    has_ov = p.carry_bits % p.carry_sub_bits != 0
    keys = []
    for t in range(T):
        st = ost[t * msb:(t + 1) * msb]
        pos = 0
        for _ in range(2 * L):
            pos += p.LB
            keys.extend(int(v) for v in st[pos:pos + 8]); pos += 8
        pos += 2 * L * L * p.WB + L * p.WB
        for c in range(C):
            pos += per_col
            if c < C - 1:
                pos += p.CB
                for j in range(p.carry_nsub):
                    ov = has_ov and j == p.carry_nsub - 1
                    keys.append((hist_tab2_off if ov else hist_tab1_off) + int(st[pos])); pos += 1
            pos += 2
    return np.array(keys)


@pytest.mark.parametrize("w,bits,e", [
    (64, 2048, 65537),                 # 20,330 cells: LDS-staged form
    (64, 2048, (1 << 33) - 1),         # 66 records, 70,620 cells: direct form
    (32, 1024, 65537),                 # 32-bit limbs: 4-bit limb sub-limbs, 5-bit carry sub-limbs (+ overflow)
])
def test_lookup_permutation(H, w, bits, e):
    """Grouped arrangement of the lookup inputs: a stable counting sort of every element's
    sub-limb cells by table row, checked against numpy's stable argsort of the oracle's sub-limb sequence."""
    chip = H.BigIntChip(w, bits)
    o = Oracle(w, bits // w)
    rng = random.Random(17)
    N = [rand_modulus(rng, bits) for _ in range(3)]
    X = [rng.randrange(n) for n in N]
    res = chip.pow_mod_fixed_exp(chip.assign_integer(X), e, chip.assign_integer(N))
    perm, rows = res.trace.lookup_permutation()
    hist_dev = res.trace.lookup_hist()
    p2, r2, h2 = res.trace.lookup_permutation(with_hist=True)   # one launch: the multiplicities are the counting pass's by-product
    assert torch.equal(p2, perm) and torch.equal(r2, rows) and torch.equal(h2, hist_dev)
    hist = hist_dev.cpu().numpy()
    perm, rows = perm.cpu().numpy(), rows.cpu().numpy()
    T = e.bit_length() + bin(e).count("1")
    # table row offsets as h2r_ctx_create lays them out: limb table, carry table (if its width differs), overflow table
    q = o.p
    tab0 = 1 << q.limb_sub_bits
    same = q.carry_sub_bits == q.limb_sub_bits
    tab1 = 0 if same else tab0
    tab2 = tab0 + (0 if same else 1 << q.carry_sub_bits)
    ovb = q.carry_bits % q.carry_sub_bits
    hist_len = tab2 + ((1 << ovb) if ovb else 0)
    assert hist.shape[1] == hist_len
    if w == 64:
        assert (tab1, tab2, hist_len) == (0, 256, 320)
    for i in range(3):
        rc, oo, ost = o.pow_mod_fixed_exp(o.limbs(X[i]), o.limbs(N[i]), e)
        keys = _oracle_lookup_keys(o, ost, T, tab1, tab2)
        assert len(keys) == perm.shape[1]
        if (w, e) == (64, 65537):
            assert len(keys) == 20330
        want = np.argsort(keys, kind="stable")
        assert np.array_equal(perm[i], want)
        assert np.array_equal(rows[i], keys[want])
        assert np.array_equal(np.bincount(keys, minlength=hist_len), hist[i])


def test_pipelined_calls_match_oracle(H):
    """h2r_pipeline_*: four back-to-back pipelined modpow batches (chain k+1 overlapping trace k) produce the
    same bytes as the oracle for every batch."""
    chip = H.BigIntChip(64, 2048)
    o = Oracle(64, 32)
    pipe = chip.pipeline()
    pl = chip.pow_fixed_layout(65537)
    rng = random.Random(31)
    B = 96
    sets = []
    for k in range(4):
        N = [rand_modulus(rng, 2048) for _ in range(B)]
        X = [rng.randrange(n) for n in N]
        if k == 2:
            X[5] = N[5] + 1      # not in field -> status, other elements unaffected
        bufs = dict(N=N, X=X, n=chip.assign_integer(N), x=chip.assign_integer(X),
                    trace=torch.empty(B * pl.elem_stride, dtype=torch.uint8, device="cuda"),
                    ws=torch.empty(chip.workspace_bytes(B, pl.num_mul_mods), dtype=torch.uint8, device="cuda"),
                    out=torch.empty((B, 32), dtype=torch.int64, device="cuda"),
                    status=torch.zeros(B, dtype=torch.uint8, device="cuda"))
        sets.append(bufs)
    for s in sets:
        pipe.modpow_public_key(s["x"], 65537, s["n"], s["trace"], s["ws"], s["out"], s["status"])
    pipe.join()
    torch.cuda.synchronize()
    for k, s in enumerate(sets):
        st = s["status"].cpu().tolist()
        assert all(v == 0 for i, v in enumerate(st) if not (k == 2 and i == 5))
        if k == 2:
            assert st[5] == H.H2R_E_NOT_IN_FIELD
        out = H.AssignedInteger(s["out"], 64).to_big_uint()
        tr = H.Trace(chip, s["trace"], B, pl)
        for i in (0, 5, 41, B - 1):
            if k == 2 and i == 5:
                continue
            assert out[i] == pow(s["X"][i], 65537, s["N"][i])
            rc, oo, ost = o.pow_mod_fixed_exp(o.limbs(s["X"][i]), o.limbs(s["N"][i]), 65537)
            assert np.array_equal(ost, tr.flatten(i)), (k, i)
    pipe.close()


@pytest.mark.parametrize("w,bits", [(64, 2048), (64, 1024), (64, 3072), (64, 4096), (32, 4096)])
def test_pipeline_one_launch_steps(H, w, bits):
    """RSA-2048 / RSA-1024 pipelined calls of 513..4,096 signatures are issued as one launch per call (step_kernel: this call's
    chains and in-field witness + the previous call's records).  A train of such calls, interrupted by a small call (the
    two-queue form) and by a change of the caller's stream, leaves byte-for-byte what the plain export writes -- trace,
    in-field witness, results, status -- and the launches are the expected ones."""
    from halo2_rsa_amd import _lib
    chip = H.BigIntChip(w, bits)
    pl = chip.pow_fixed_layout(65537)
    ies = chip.in_field_layout()[0]
    rng = random.Random(77)
    sizes = [640, 640, 128, 640, 640, 640]
    sets = []
    for k, B in enumerate(sizes):
        N = [rand_modulus(rng, bits) for _ in range(B)]
        X = [rng.randrange(n) for n in N]
        if k == 1:
            X[7] = N[7] + 5      # not in field: status, no records for that element, its in-field witness still written
        mk = lambda nbytes: torch.zeros(nbytes, dtype=torch.uint8, device="cuda")
        sets.append(dict(B=B, N=N, X=X, n=chip.assign_integer(N), x=chip.assign_integer(X),
                         trace=mk(B * pl.elem_stride), inf=mk(B * ies), ws=mk(chip.workspace_bytes(B, pl.num_mul_mods)),
                         out=torch.zeros((B, bits // w), dtype=chip.torch_dtype, device="cuda"), status=mk(B),
                         ref_trace=mk(B * pl.elem_stride), ref_inf=mk(B * ies)))
    torch.cuda.synchronize()
    pipe = chip.pipeline()
    other = torch.cuda.Stream()
    _lib.profile_enable(64)
    for k, s in enumerate(sets):
        if k == 4:
            other.wait_stream(torch.cuda.current_stream())
        ctx = torch.cuda.stream(other) if k >= 4 else torch.cuda.stream(torch.cuda.current_stream())
        with ctx:
            pipe.modpow_public_key(s["x"], 65537, s["n"], s["trace"], s["ws"], s["out"], s["status"], in_field_buf=s["inf"])
    with torch.cuda.stream(other):
        pipe.join()
    torch.cuda.synchronize()
    n_step, n_chain, n_rec = (len(_lib.profile_read(k)) for k in (_lib.KERNEL_STEP, _lib.KERNEL_CHAIN, _lib.KERNEL_TRACE))
    _lib.profile_enable(0)
    # calls 0,1 | small call 2 | call 3 | stream change | calls 4,5:  steps = (1 carries 0's records) + (5 carries 4's)
    assert n_step == 2, n_step
    assert n_chain == 4, n_chain      # calls 0, 2, 3, 4 start with a chain kernel of their own
    assert n_rec == 4, n_rec          # records of calls 1, 2, 3, 5 written by the record kernel alone
    for k, s in enumerate(sets):
        ref = chip.pow_mod_fixed_exp(s["x"], 65537, s["n"], trace_buf=s["ref_trace"], check_in_field=True, in_field_buf=s["ref_inf"])
        torch.cuda.synchronize()
        assert torch.equal(ref.status, s["status"]), k
        ok = ref.status == 0     # (an element with a status gets no result row: compare the rows that exist)
        assert torch.equal(ref.value.limbs_dev[ok], s["out"][ok]), k
        assert torch.equal(s["ref_trace"], s["trace"]), k
        assert torch.equal(s["ref_inf"], s["inf"]), k
        st = s["status"].cpu().tolist()
        assert all(v == 0 for i, v in enumerate(st) if not (k == 1 and i == 7))
        if k == 1:
            assert st[7] == H.H2R_E_NOT_IN_FIELD
    pipe.close()


@pytest.mark.parametrize("depth,side_streams,toggle_profiler", [(2, 1, False), (2, 2, False), (3, 2, False), (4, 1, False),
                                                              (2, 1, True), (3, 2, True)])
def test_pipeline_buffer_rotation(H, depth, side_streams, toggle_profiler):
    """h2r_pipeline_create_ex: seven calls rotating through `depth` buffer sets.  The contract under test: when call
    k returns, the caller's stream is ordered after the records of call k - depth + 1, so the set about to be reused
    can be copied out (stream-ordered) right before the next call overwrites it.  Every copy must equal the oracle.
    toggle_profiler: the per-kernel event profiler is armed, released and exhausted between calls -- the pipeline
    borrows the profiler's stop events while it is armed and must survive their release."""
    from halo2_rsa_amd import _lib
    chip = H.BigIntChip(64, 2048)
    o = Oracle(64, 32)
    pipe = H.Pipeline(chip, depth=depth, side_streams=side_streams)
    pl = chip.pow_fixed_layout(65537)
    rng = random.Random(77 + depth)
    B, CALLS = 64, 7
    sets = [dict(trace=torch.empty(B * pl.elem_stride, dtype=torch.uint8, device="cuda"),
                 ws=torch.empty(chip.workspace_bytes(B, pl.num_mul_mods), dtype=torch.uint8, device="cuda"),
                 out=torch.empty((B, 32), dtype=torch.int64, device="cuda"),
                 status=torch.zeros(B, dtype=torch.uint8, device="cuda")) for _ in range(depth)]
    inputs, snaps = [], {}
    for k in range(CALLS):
        N = [rand_modulus(rng, 2048) for _ in range(B)]
        X = [rng.randrange(n) for n in N]
        inputs.append((N, X, chip.assign_integer(N), chip.assign_integer(X)))
        s = sets[k % depth]
        if k >= depth:   # copy call k-depth's results out before its buffers are reused
            snaps[k - depth] = (s["trace"].clone(), s["out"].clone(), s["status"].clone())
        if toggle_profiler:
            if k == 1:
                _lib.profile_enable(64)
            if k == 3:
                _lib.profile_enable(0)      # releases the events calls 1 and 2 lent to the pipeline
            if k == 4:
                _lib.profile_enable(3)      # runs out of capacity in the middle of call 5
        pipe.modpow_public_key(inputs[k][3], 65537, inputs[k][2], s["trace"], s["ws"], s["out"], s["status"])
    if toggle_profiler:
        _lib.profile_enable(0)
    pipe.join()
    for k in range(CALLS - depth, CALLS):
        s = sets[k % depth]
        snaps[k] = (s["trace"].clone(), s["out"].clone(), s["status"].clone())
    torch.cuda.synchronize()
    for k in range(CALLS):
        N, X = inputs[k][0], inputs[k][1]
        trace, out, status = snaps[k]
        assert not status.cpu().numpy().any()
        got = H.AssignedInteger(out, 64).to_big_uint()
        assert all(got[i] == pow(X[i], 65537, N[i]) for i in range(B)), k
        tr = H.Trace(chip, trace, B, pl)
        for i in (0, 17, B - 1):
            rc, oo, ost = o.pow_mod_fixed_exp(o.limbs(X[i]), o.limbs(N[i]), 65537)
            assert np.array_equal(ost, tr.flatten(i)), (k, i)
    pipe.close()
    # shape errors where the C ABI defines them
    bad = ctypes.c_void_p()
    assert H.lib().h2r_pipeline_create_ex(chip._ctx, 1, 1, ctypes.byref(bad)) == H.H2R_E_SHAPE
    assert H.lib().h2r_pipeline_create_ex(chip._ctx, 2, 3, ctypes.byref(bad)) == H.H2R_E_SHAPE


@pytest.mark.parametrize("depth,side_streams", [(2, 1), (3, 2)])
def test_pipeline_full_size_rotation(H, depth, side_streams):
    """The rotation contract at BASELINE config 2's size (1,024 signatures per call: the record kernel of call k really
    overlaps the chain kernel of call k+1 and, with two record streams, the record kernel of call k+1).  Five calls with
    different inputs over `depth` buffer sets; each call's results are copied out right before its buffers are reused
    and must equal pow(x, e, n) for every element, with byte-exact traces for sampled elements."""
    chip = H.BigIntChip(64, 2048)
    o = Oracle(64, 32)
    pipe = H.Pipeline(chip, depth=depth, side_streams=side_streams)
    pl = chip.pow_fixed_layout(65537)
    rng = random.Random(4242 + depth)
    B, CALLS = 1024, 5
    sets = [dict(trace=torch.empty(B * pl.elem_stride, dtype=torch.uint8, device="cuda"),
                 ws=torch.empty(chip.workspace_bytes(B, pl.num_mul_mods), dtype=torch.uint8, device="cuda"),
                 out=torch.empty((B, 32), dtype=torch.int64, device="cuda"),
                 status=torch.zeros(B, dtype=torch.uint8, device="cuda")) for _ in range(depth)]
    base_n = [rand_modulus(rng, 2048) for _ in range(B)]
    inputs, snaps = [], {}
    for k in range(CALLS):
        N = base_n[k:] + base_n[:k]                      # a different pairing of moduli and bases per call
        X = [(n >> (k + 1)) ^ (0x9e3779b97f4a7c15 * (i + 1) * (k + 1)) for i, n in enumerate(N)]
        X = [x % n for x, n in zip(X, N)]
        inputs.append((N, X, chip.assign_integer(N), chip.assign_integer(X)))
    audits = {}

    def audit(k):   # EVERY record of call k, in place (its trace is complete and its workspace intact at this point)
        s = sets[k % depth]
        res = H.BatchResult(None, H.Trace(chip, s["trace"], B, pl), s["status"], None, s["ws"],
                            ("pow_fixed", inputs[k][3], None, inputs[k][2], (65537).to_bytes(3, "little")))
        audits[k] = res.audit()[0]
    for k in range(CALLS):
        s = sets[k % depth]
        if k >= depth:
            audit(k - depth)
            snaps[k - depth] = (s["trace"].clone(), s["out"].clone(), s["status"].clone())
        pipe.modpow_public_key(inputs[k][3], 65537, inputs[k][2], s["trace"], s["ws"], s["out"], s["status"])
    pipe.join()
    for k in range(CALLS - depth, CALLS):
        s = sets[k % depth]
        audit(k)
        snaps[k] = (s["trace"].clone(), s["out"].clone(), s["status"].clone())
    torch.cuda.synchronize()
    for k in range(CALLS):
        N, X = inputs[k][0], inputs[k][1]
        trace, out, status = snaps[k]
        assert not status.cpu().numpy().any()
        assert not audits[k].cpu().numpy().any(), k      # the bench's exact path (step launches, 1,024 per call), every record
        got = H.AssignedInteger(out, 64).to_big_uint()
        assert all(got[i] == pow(X[i], 65537, N[i]) for i in range(B)), k
        tr = H.Trace(chip, trace, B, pl)
        for i in (0, 511, B - 1):
            rc, oo, ost = o.pow_mod_fixed_exp(o.limbs(X[i]), o.limbs(N[i]), 65537)
            assert np.array_equal(ost, tr.flatten(i)), (k, i)
    pipe.close()


@pytest.mark.parametrize("B", [1280, 4096, 6400])
def test_pipeline_two_queue_rsa1024(H, B):
    """[r6] RSA-1024 (the reference's enabled bench size, benches/bench.rs:393-407) calls of 1,280 signatures and more (above 4,096: walked as
    sub-batches of 2,048 -- 6,400 ends with a ragged one) on a pipeline with two side
    streams and three buffer sets take the two-queue form when the streams sit on three hardware queues (one-wave chain kernels on the caller's
    stream, record kernels alternating between the side streams), else the one-launch step.  Either way five rotating calls leave, byte for byte,
    what the plain export writes: every record of every call (audited in place before its set is reused), results, statuses."""
    from halo2_rsa_amd import _lib
    chip = H.BigIntChip(64, 1024)
    depth = 3
    pipe = H.Pipeline(chip, depth=depth, side_streams=2)
    info = pipe.info(B)
    assert info.three_queues in (0, 1) and (info.record_form == _lib.H2R_PIPE_TWO_QUEUE) == (info.three_queues == 1)
    assert pipe.info(1024).three_queues == 2 and pipe.info(1024).record_form == _lib.H2R_PIPE_ONE_LAUNCH_STEP     # at 1,024 per call the step stays
    pl = chip.pow_fixed_layout(65537)
    rng = random.Random(1024 + B)
    CALLS = 5
    mk = lambda nbytes: torch.zeros(nbytes, dtype=torch.uint8, device="cuda")   # (zeros: the records' padding is never written)
    sets = [dict(trace=mk(B * pl.elem_stride), ws=mk(chip.workspace_bytes(B, pl.num_mul_mods)),
                 out=torch.empty((B, 16), dtype=torch.int64, device="cuda"), status=torch.zeros(B, dtype=torch.uint8, device="cuda")) for _ in range(depth)]
    base_n = [rand_modulus(rng, 1024) for _ in range(256)]
    inputs, outs, audits, digests = [], {}, {}, {}
    for k in range(CALLS):
        N = [base_n[(i * 7 + k) % 256] for i in range(B)]
        X = [((n >> (k + 1)) ^ (0x9e3779b97f4a7c15 * (i + 1) * (k + 1))) % n for i, n in enumerate(N)]
        if k == 3:
            X[11] = N[11] + 1                                  # not in field: a status, no records for that element
        inputs.append((N, X, chip.assign_integer(N), chip.assign_integer(X)))

    def leave(k):   # call k's set right before it is reused: every record audited in place, results and a digest of the trace kept
        s = sets[k % depth]
        res = H.BatchResult(None, H.Trace(chip, s["trace"], B, pl), s["status"], None, s["ws"],
                            ("pow_fixed", inputs[k][3], None, inputs[k][2], (65537).to_bytes(3, "little")))
        audits[k] = res.audit()[0]
        outs[k] = (s["out"].clone(), s["status"].clone())
        digests[k] = s["trace"].view(torch.int64).sum()        # (stream-ordered; compared with the plain export's below)
    _lib.profile_enable(64)
    for k in range(CALLS):
        if k >= depth:
            leave(k - depth)
        s = sets[k % depth]
        pipe.modpow_public_key(inputs[k][3], 65537, inputs[k][2], s["trace"], s["ws"], s["out"], s["status"])
    pipe.join()
    torch.cuda.synchronize()
    n_step, n_rec = len(_lib.profile_read(_lib.KERNEL_STEP)), len(_lib.profile_read(_lib.KERNEL_TRACE))
    _lib.profile_enable(0)
    if info.three_queues == 1:
        assert n_step == 0 and n_rec >= CALLS, (n_step, n_rec)  # record kernels of their own, no step launch
    else:
        assert n_step >= CALLS - 1, (n_step, n_rec)
    for k in range(CALLS - depth, CALLS):
        leave(k)
    ref_trace = mk(B * pl.elem_stride)
    for k in range(CALLS):
        N, X = inputs[k][0], inputs[k][1]
        out, status = outs[k]
        st = status.cpu().tolist()
        assert all(v == (H.H2R_E_NOT_IN_FIELD if (k == 3 and i == 11) else 0) for i, v in enumerate(st)), k
        bad = audits[k].cpu().numpy()
        assert not np.delete(bad, 11 if k == 3 else []).any(), k
        got = H.AssignedInteger(out, 64).to_big_uint()
        assert all(got[i] == pow(X[i], 65537, N[i]) for i in range(B) if not (k == 3 and i == 11)), k
        ref_trace.zero_()
        ref = chip.pow_mod_fixed_exp(inputs[k][3], 65537, inputs[k][2], trace_buf=ref_trace, check_in_field=True)
        assert torch.equal(ref.status, status), k
        if k >= CALLS - depth:                                 # the sets still hold these calls: byte for byte
            tr = sets[k % depth]["trace"]
            if k == 3:                                         # (the failed element's trace is unspecified)
                a, b = tr.view(B, -1), ref_trace.view(B, -1)
                keep = torch.ones(B, dtype=torch.bool, device="cuda"); keep[11] = False
                assert torch.equal(a[keep], b[keep]), k
            else:
                assert torch.equal(tr, ref_trace), k
        elif k != 3:
            assert int(digests[k].item()) == int(ref_trace.view(torch.int64).sum().item()), k
    pipe.close()


@pytest.mark.parametrize("B", [640, 96])
def test_pipeline_inputs_may_be_refilled_between_calls(H, B):
    """include/h2r.h: the inputs of a pipelined call are read in stream order INSIDE the call.  A producer with ONE x and ONE n
    staging buffer refills them (stream-ordered copies) as soon as a call has returned; every call's in-field witness, records
    and results must still be those of ITS inputs -- for the one-launch-step form (640 per call: round 2 computed call k's
    in-field witness in call k+1's launch, i.e. from call k+1's x) and for the two-queue form (96 per call)."""
    chip = H.BigIntChip(64, 2048)
    pl = chip.pow_fixed_layout(65537)
    ies = chip.in_field_layout()[0]
    rng = random.Random(991 + B)
    CALLS = 4
    mk = lambda nbytes: torch.zeros(nbytes, dtype=torch.uint8, device="cuda")
    sets = [dict(trace=mk(B * pl.elem_stride), inf=mk(B * ies), ws=mk(chip.workspace_bytes(B, pl.num_mul_mods)),
                 out=torch.zeros((B, 32), dtype=torch.int64, device="cuda"), status=mk(B)) for _ in range(CALLS)]
    inputs = []
    for k in range(CALLS):
        N = [rand_modulus(rng, 2048) for _ in range(B)]
        X = [rng.randrange(n) for n in N]
        inputs.append((chip.assign_integer(N), chip.assign_integer(X)))
    x_stage = chip.assign_integer([0] * B)
    n_stage = chip.assign_integer([1] * B)
    torch.cuda.synchronize()
    pipe = chip.pipeline()
    for k, s in enumerate(sets):
        n_stage.limbs_dev.copy_(inputs[k][0].limbs_dev)      # refilled in stream order right after the previous call returned
        x_stage.limbs_dev.copy_(inputs[k][1].limbs_dev)
        pipe.modpow_public_key(x_stage, 65537, n_stage, s["trace"], s["ws"], s["out"], s["status"], in_field_buf=s["inf"])
    n_stage.limbs_dev.zero_()
    x_stage.limbs_dev.zero_()
    pipe.join()
    torch.cuda.synchronize()
    for k, s in enumerate(sets):
        ref_trace, ref_inf = mk(B * pl.elem_stride), mk(B * ies)
        ref = chip.pow_mod_fixed_exp(inputs[k][1], 65537, inputs[k][0], trace_buf=ref_trace, check_in_field=True, in_field_buf=ref_inf)
        torch.cuda.synchronize()
        assert not s["status"].cpu().numpy().any(), k
        assert torch.equal(ref.value.limbs_dev, s["out"]), k
        assert torch.equal(ref_inf, s["inf"]), k
        assert torch.equal(ref_trace, s["trace"]), k
    pipe.close()


@pytest.mark.parametrize("shared", [False, True])
def test_pipeline_call_walked_as_sub_batches(H, shared):
    """A large pipelined call that finds the pipeline empty (3,840 RSA-2048 signatures: 1,024 + 1,536 + 1,280) is walked by
    the library as sub-batches with their own chain and record kernels; the same call behind a record kernel in flight is
    one launch.  The caller must not be able to tell:
    trace, in-field witness, results and statuses equal the one-call export's byte for byte, the workspace is the whole
    call's plan (the in-place audit of every record runs on it), also with one shared modulus."""
    from halo2_rsa_amd import big_integer as BI
    chip = H.BigIntChip(64, 2048)
    pl = chip.pow_fixed_layout(65537)
    rng = random.Random(909 + shared)
    B = 3840
    base = [rand_modulus(rng, 2048) for _ in range(48)]
    N = [base[0]] if shared else [base[i % 48] ^ ((i // 48) << 70) | 1 for i in range(B)]
    X = [((base[i % 48] >> 3) * (2 * i + 1) + i) % N[0 if shared else i] for i in range(B)]
    if not shared:
        X[1500] = N[1500] + 5       # not in field: status only, in the second sub-batch
    x, n = chip.assign_integer(X), chip.assign_integer(N)
    ref = chip.pow_mod_fixed_exp(x, 65537, n, check_in_field=True,
                                 trace_buf=torch.zeros(B * pl.elem_stride, dtype=torch.uint8, device="cuda"))
    sets = [dict(trace=torch.zeros(B * pl.elem_stride, dtype=torch.uint8, device="cuda"),
                 ws=torch.zeros(chip.workspace_bytes(B, pl.num_mul_mods), dtype=torch.uint8, device="cuda"),
                 out=torch.zeros((B, 32), dtype=torch.int64, device="cuda"),
                 status=torch.zeros(B, dtype=torch.uint8, device="cuda"),
                 inf=torch.zeros(B * chip.in_field_layout()[0], dtype=torch.uint8, device="cuda")) for _ in range(2)]
    pipe = chip.pipeline()
    torch.cuda.synchronize()
    for b in sets:   # the first call finds the pipeline empty (sub-batches), the second one the first call's record kernels
        pipe.modpow_public_key(x, 65537, n, b["trace"], b["ws"], b["out"], b["status"], in_field_buf=b["inf"])
    pipe.join()
    torch.cuda.synchronize()
    for b in sets:
        status, out = b["status"], b["out"]
        assert torch.equal(status, ref.status)
        st = status.cpu().numpy()
        assert (st != 0).sum() == (0 if shared else 1) and (shared or st[1500] == H.H2R_E_NOT_IN_FIELD)
        ok = torch.from_numpy(st == 0).cuda()      # the element with a status has no defined result or records
        assert torch.equal(out[ok], ref.value.limbs_dev[ok])
        got = H.AssignedInteger(out, 64).to_big_uint()
        assert all(got[i] == pow(X[i], 65537, N[0 if shared else i]) for i in range(B) if st[i] == 0)
        for lo_ in range(0, B, 512):                 # every byte of every element's records
            sl = slice(lo_, lo_ + 512)
            assert torch.equal(b["trace"].view(B, -1)[sl][ok[sl]], ref.trace.buf.view(B, -1)[sl][ok[sl]]), lo_
        assert torch.equal(b["inf"].view(B, -1)[ok], ref.in_field.buf[:b["inf"].numel()].view(B, -1)[ok])
        tr = H.Trace(chip, b["trace"], B, pl)
        res = BI.BatchResult(H.AssignedInteger(out, 64), tr, status, workspace=b["ws"], inputs=ref.inputs)
        bad, first = res.audit()
        torch.cuda.synchronize()
        nb = bad.cpu().numpy()
        assert not nb[st == 0].any(), ("audit", int(np.nonzero(nb)[0][0]))
    pipe.close()


def test_plain_large_calls_from_two_threads(H):
    """The stream-ordered export walks a large call through a pipeline the ctx owns (overlapped_pow_fixed).  Two host
    threads, one ctx, their own streams, interleaved calls of 2,304 signatures: every call's results and every record must
    be what a small-batch call gives (the in-place audit runs on the caller's stream right behind the call)."""
    import threading
    chip = H.BigIntChip(64, 2048)
    rng = random.Random(2626)
    B = 2304
    base = [rand_modulus(rng, 2048) for _ in range(32)]
    errors = []

    def worker(seed):
        try:
            st = torch.cuda.Stream()
            with torch.cuda.stream(st):
                for it in range(3):
                    N = [base[(i + seed + it) % 32] ^ ((i // 32 + seed) << 80) | 1 for i in range(B)]
                    X = [((base[(i * 7 + it) % 32] >> 5) * (2 * i + 3 + seed)) % N[i] for i in range(B)]
                    res = chip.pow_mod_fixed_exp(chip.assign_integer(X), 65537, chip.assign_integer(N), check_in_field=True)
                    bad, first = res.audit()
                    st.synchronize()
                    assert not res.status.cpu().numpy().any()
                    got = res.value.to_big_uint()
                    assert all(got[i] == pow(X[i], 65537, N[i]) for i in range(B)), (seed, it)
                    assert not bad.cpu().numpy().any(), (seed, it)
        except Exception as ex:   # surfaced in the main thread
            errors.append((seed, repr(ex)))

    threads = [threading.Thread(target=worker, args=(k,)) for k in (1, 2)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors


def test_pipeline_with_device_side_consumers(H):
    """The use the pipeline is built for: while call k+1 is in flight, the caller's stream consumes call k-1's trace with
    the device-side emitter and the advice-image kernel (both LDS-heavy kernels queued between pipelined calls) and audits
    it in place.  Six calls over two buffer sets, 768 signatures each; every emitted stream must equal the oracle's for
    sampled elements and the host walk for all, every audit must be clean."""
    from halo2_rsa_amd import big_integer as BI
    chip = H.BigIntChip(64, 2048)
    o = Oracle(64, 32)
    pl = chip.pow_fixed_layout(65537)
    rng = random.Random(5151)
    B, CALLS, depth = 768, 6, 2
    base = [rand_modulus(rng, 2048) for _ in range(64)]
    pipe = H.Pipeline(chip, depth=depth, side_streams=1)
    sets = [dict(trace=torch.zeros(B * pl.elem_stride, dtype=torch.uint8, device="cuda"),
                 ws=torch.zeros(chip.workspace_bytes(B, pl.num_mul_mods), dtype=torch.uint8, device="cuda"),
                 out=torch.zeros((B, 32), dtype=torch.int64, device="cuda"),
                 status=torch.zeros(B, dtype=torch.uint8, device="cuda")) for _ in range(depth)]
    inputs, consumed = [], {}

    def consume(k):   # everything here is queued on the caller's stream, ordered by the pipeline's contract
        s = sets[k % depth]
        tr = H.Trace(chip, s["trace"], B, pl)
        res = BI.BatchResult(H.AssignedInteger(s["out"], 64), tr, s["status"], workspace=s["ws"],
                             inputs=("pow_fixed", inputs[k][3], None, inputs[k][2], b"\x01\x00\x01"))
        stream = tr.emit_stream()
        img = res.emit_advice()
        bad, first = res.audit()
        consumed[k] = (stream, img[:4].clone(), bad, s["out"].clone(), s["status"].clone())

    for k in range(CALLS):
        N = [base[(i + 5 * k) % 64] ^ ((i // 64 + k) << 100) | 1 for i in range(B)]
        X = [((base[(i * 3 + k) % 64] >> 9) * (i + 11 + k)) % N[i] for i in range(B)]
        inputs.append((N, X, chip.assign_integer(N), chip.assign_integer(X)))
        s = sets[k % depth]
        pipe.modpow_public_key(inputs[k][3], 65537, inputs[k][2], s["trace"], s["ws"], s["out"], s["status"])
        if k >= 1:
            consume(k - 1)    # call k has returned: the stream is ordered after call k-1's records
    pipe.join()
    consume(CALLS - 1)
    torch.cuda.synchronize()
    for k in range(CALLS):
        N, X = inputs[k][0], inputs[k][1]
        stream, img, bad, out, status = consumed[k]
        assert not status.cpu().numpy().any() and not bad.cpu().numpy().any(), k
        got = H.AssignedInteger(out, 64).to_big_uint()
        assert all(got[i] == pow(X[i], 65537, N[i]) for i in range(B)), k
        host = stream.cpu().numpy()
        for i in (0, 383, B - 1):
            rc, oo, ost = o.pow_mod_fixed_exp(o.limbs(X[i]), o.limbs(N[i]), 65537)
            assert rc == 0 and np.array_equal(host[i, :len(ost)], ost), (k, i)
    pipe.close()


def test_trace_arena(H):
    """h2r_arena_create: candidate regions are mapped and measured, the fastest kept (sorted), the others given back; the
    kept regions are ordinary device memory -- a pipelined modpow_public_key into them is byte-exact against the oracle and
    audits clean -- and a second arena can be created after the first one is destroyed.  Shape errors where the C ABI
    defines them."""
    from halo2_rsa_amd import big_integer as BI
    chip = H.BigIntChip(64, 2048)
    o = Oracle(64, 32)
    pl = chip.pow_fixed_layout(65537)
    B = 256
    arena = H.TraceArena.for_pow(chip, 65537, B, regions=2, candidates=5)
    # 5 candidates, then further ones while the kept regions are not of one fast class (up to 3 x 5 one at a time, then up to four
    # more rounds of 5 behind placeholders)
    assert len(arena.regions) == 2 and 5 <= len(arena.measurements_ms) <= 5 + 15 + 20
    assert all(t > 0 for t in arena.measurements_ms)
    assert arena.region_ms == sorted(arena.measurements_ms)[:2]
    assert all(r.numel() == B * pl.elem_stride and r.is_cuda for r in arena.regions)
    rng = random.Random(808)
    N = [rand_modulus(rng, 2048) for _ in range(B)]
    X = [rng.randrange(n) for n in N]
    x, n = chip.assign_integer(X), chip.assign_integer(N)
    pipe = chip.pipeline()
    outs = []
    for r in arena.regions:
        ws = torch.zeros(chip.workspace_bytes(B, pl.num_mul_mods), dtype=torch.uint8, device="cuda")
        out = torch.zeros((B, 32), dtype=torch.int64, device="cuda")
        status = torch.zeros(B, dtype=torch.uint8, device="cuda")
        pipe.modpow_public_key(x, 65537, n, r, ws, out, status)
        outs.append((ws, out, status))
    pipe.join()
    torch.cuda.synchronize()
    for r, (ws, out, status) in zip(arena.regions, outs):
        assert not status.cpu().numpy().any()
        tr = H.Trace(chip, r, B, pl)
        for i in (0, 100, B - 1):
            rc, oo, ost = o.pow_mod_fixed_exp(o.limbs(X[i]), o.limbs(N[i]), 65537)
            assert rc == 0 and np.array_equal(ost, tr.flatten(i)), i
        res = BI.BatchResult(H.AssignedInteger(out, 64), tr, status, workspace=ws, inputs=("pow_fixed", x, None, n, b"\x01\x00\x01"))
        bad, first = res.audit()
        torch.cuda.synchronize()
        assert not bad.cpu().numpy().any()
    pipe.close()
    arena.close()
    again = H.TraceArena.for_pow(chip, 65537, 64, regions=1, candidates=2)
    assert len(again.regions) == 1
    again.close()
    a = ctypes.c_void_p()
    L = H.lib()
    assert L.h2r_arena_create(chip._ctx, pl.elem_stride, pl.off_records, pl.num_mul_mods, 16, 3, 2, None, ctypes.byref(a)) == H.H2R_E_SHAPE   # fewer candidates than regions
    assert L.h2r_arena_create(chip._ctx, 1024, pl.off_records, pl.num_mul_mods, 16, 1, 2, None, ctypes.byref(a)) == H.H2R_E_SHAPE              # element too small for its records
    assert L.h2r_arena_create(None, pl.elem_stride, pl.off_records, pl.num_mul_mods, 16, 1, 2, None, ctypes.byref(a)) == 7   # H2R_E_NULL


@pytest.mark.parametrize("w,L,batch", [(64, 32, 433), (64, 48, 217), (32, 128, 109), (64, 16, 869)])
def test_xcd_mapping_with_ragged_grids(H, w, L, batch):
    """The record kernel, the emitter and the advice kernel give every XCD a contiguous eighth of a launch's workgroups
    (xcd_contiguous_block, launches of >= 2,048 workgroups).  Batches whose workgroup count is NOT a multiple of 8 and whose
    last workgroup is partly empty: every result, sampled streams (host walk and device emitter) against the oracle, every
    record audited in place, and the advice image of the last element against the Python restatement's row count."""
    chip = H.BigIntChip(w, w * L)
    o = Oracle(w, L)
    rng = random.Random(3 * w + L + batch)
    base = [rand_modulus(rng, w * L) for _ in range(16)]
    N = [base[i % 16] ^ ((i // 16) << 40) | 1 for i in range(batch)]
    X = [((base[(i * 5) % 16] >> 7) * (i + 3)) % N[i] for i in range(batch)]
    res = chip.pow_mod_fixed_exp(chip.assign_integer(X), 65537, chip.assign_integer(N))
    _check_pow_batch(H, chip, o, X, N, 65537, res, [0, 1, batch // 2, batch - 2, batch - 1], rng)
    stream = res.trace.emit_stream()
    torch.cuda.synchronize()
    host = stream.cpu().numpy()
    for i in (0, batch // 3, batch - 1):
        rc, oo, ost = o.pow_mod_fixed_exp(o.limbs(X[i]), o.limbs(N[i]), 65537)
        assert rc == 0 and np.array_equal(host[i, :len(ost)], ost), i
    img = res.emit_advice()
    torch.cuda.synchronize()
    rows = int(H.lib().h2r_advice_rows(chip._ctx))
    assert img.shape == (batch, (2 + 19 * rows) * 160)          # two constant rows (acc = 1), then the 19 records
    last = img[batch - 1].cpu().numpy().reshape(2 + 19 * rows, 5, 32)
    first = img[0].cpu().numpy().reshape(2 + 19 * rows, 5, 32)
    for im, i in ((first, 0), (last, batch - 1)):   # row 0 of record 0 holds the first four sub-limbs of q[0] and, in column e, q[0]
        assert int.from_bytes(im[0, 0].tobytes(), "little") == 1 and not im[0, 1:].any() and not im[1].any()
        q0 = (X[i] * X[i]) // N[i] & ((1 << w) - 1)
        sb = w // 8
        subs = [(q0 >> (sb * k)) & ((1 << sb) - 1) for k in range(4)]
        assert [int.from_bytes(im[2, k].tobytes(), "little") for k in range(4)] == subs, i
        assert int.from_bytes(im[2, 4].tobytes(), "little") == q0, i


def _check_pow_batch(H, chip, o, X, N, e, res, sample, rng):
    torch.cuda.synchronize()
    assert not res.status.cpu().numpy().any()
    out = res.value.to_big_uint()
    assert all(out[i] == pow(X[i], e, N[i]) for i in range(len(X)))
    for i in sample:
        rc, oo, ost = o.pow_mod_fixed_exp(o.limbs(X[i]), o.limbs(N[i]), e)
        assert rc == 0 and np.array_equal(ost, res.trace.flatten(i)), i
    # EVERY record of EVERY element, checked where it lies in HBM (SURVEY Appendix C invariants 1-6): sub-limbs, each
    # accumulator against its predecessor + one product, eq_b, every carry step, flags, chain linkage, result limbs
    bad, first = res.audit()
    torch.cuda.synchronize()
    nb = bad.cpu().numpy()
    assert not nb.any(), ("audit", int(np.nonzero(nb)[0][0]), hex(int(first.cpu().numpy()[np.nonzero(nb)[0][0]])))


@pytest.mark.parametrize("w,L", [(64, 32), (32, 128), (64, 48), (64, 8), (32, 24)])
def test_audit_accepts_valid_and_flags_every_corruption(H, w, L):
    """The in-place checker (h2r_*_trace_check) is itself checked: 0 violations on traces the oracle comparison accepts,
    and the expected relation code after one byte of one plane of one record is flipped (every plane class in turn)."""
    from halo2_rsa_amd import _lib
    chip = H.BigIntChip(w, w * L)
    rng = random.Random(17 * w + L)
    batch = 6
    N = [rand_modulus(rng, w * L, odd=(i != 2)) for i in range(batch)]
    X = [rng.randrange(n) for n in N]
    e = 0b100101
    res = chip.pow_mod_fixed_exp(chip.assign_integer(X), e, chip.assign_integer(N))
    bad, first = res.audit()
    torch.cuda.synchronize()
    assert not bad.cpu().numpy().any()
    mm = chip.mul_mod(chip.assign_integer(X), chip.assign_integer(X[::-1]), chip.assign_integer(N))
    bad, first = mm.audit()
    torch.cuda.synchronize()
    assert not bad.cpu().numpy().any()
    lo, pl = chip.layout, res.trace.pow_layout
    P = {n: k for k, n in enumerate(_lib.PLANES)}
    T = pl.num_mul_mods
    # (plane, entry index, expected codes): one flipped bit each, in different elements / records
    cases = [("Q", 1, {1, 2}), ("R_SUB", 0, {1}), ("AB_LO", 5, {2, 4}), ("QN_LO", 3, {2, 3, 4}), ("EQB_LO", 2, {3, 4}), ("AMB_LO", 4, {4, 5}),
             ("SUM_LO", 3, {5, 6}), ("CARRY", 2, {5, 6, 8, 9}), ("CMOD", 1, {6, 8}), ("NQ1_LO", 0, {6}), ("AMNQ1", 2, {6}),
             ("ACCX_LO", 1, {7}), ("QACC", 1, {7, 8}), ("MODACC", 0, {7, 8}), ("NQ2_LO", 2, {7}), ("AMNQ2", 3, {7}),
             ("FLAGS", 2, {8}), ("CARRY_DUP", 1, {9}), ("CARRY_SUB", 0, {9})]
    for k, (plane, idx, codes) in enumerate(cases):
        elem, t = k % batch, (k * 7) % T
        if plane in ("AB_LO", "QN_LO"):
            off = lo.plane_off[P[plane]] + idx * 16   # entry (j = 0, i = idx): the first accumulator of column idx
        else:
            off = lo.plane_off[P[plane]] + idx * lo.plane_elem[P[plane]]
        pos = elem * res.trace.elem_stride + pl.off_records + t * lo.record_stride + off
        old = int(res.trace.buf[pos].item())
        res.trace.buf[pos] = old ^ 1
        bad, first = res.audit()
        torch.cuda.synchronize()
        nb, fb = bad.cpu().numpy(), first.cpu().numpy()
        assert nb[elem] > 0 and (nb[np.arange(batch) != elem] == 0).all(), (plane, nb)
        assert (int(fb[elem]) >> 8) == t and (int(fb[elem]) & 0xff) in codes, (plane, hex(int(fb[elem])), codes)
        res.trace.buf[pos] = old
    # chain linkage: swap the result limbs of one element, then an operand in the operands buffer
    pos = 3 * res.trace.elem_stride + pl.off_result
    res.trace.buf[pos] ^= 1
    bad, first = res.audit()
    torch.cuda.synchronize()
    assert int(bad[3].item()) == 1 and (int(first[3].item()) & 0xff) == 24
    res.trace.buf[pos] ^= 1
    bad, first = res.audit()
    torch.cuda.synchronize()
    assert not bad.cpu().numpy().any()


def test_config3_shard_size_properties(H):
    """BASELINE config 3: one GPU's shard (8,192 signatures of the 65,536) -- every result equals pow(x, e, n),
    sampled elements byte-exact, and the per-element lookup multiplicities always sum to 20,330."""
    chip = H.BigIntChip(64, 2048)
    o = Oracle(64, 32)
    rng = random.Random(0x68327273 + 3)
    B = 8192
    N = [rand_modulus(rng, 2048) for _ in range(B)]
    X = [rng.randrange(n) for n in N]
    res = chip.pow_mod_fixed_exp(chip.assign_integer(X), 65537, chip.assign_integer(N))
    _check_pow_batch(H, chip, o, X, N, 65537, res, [0, 1, B // 2, B - 1] + rng.sample(range(B), 4), rng)
    hist = res.trace.lookup_hist()
    assert int(hist.sum(dim=1).min().item()) == int(hist.sum(dim=1).max().item()) == 20330


def test_config4_rsa4096_w32_full_batch(H):
    """BASELINE config 4: RSA-4096 as 128 x 32-bit limbs, batch 4096 (44 GB of trace)."""
    chip = H.BigIntChip(32, 4096)
    o = Oracle(32, 128)
    rng = random.Random(0x68327273 + 4)
    B = 4096
    N = [rand_modulus(rng, 4096) for _ in range(B)]
    X = [rng.randrange(n) for n in N]
    res = chip.pow_mod_fixed_exp(chip.assign_integer(X), 65537, chip.assign_integer(N))
    _check_pow_batch(H, chip, o, X, N, 65537, res, [0, B - 1] + rng.sample(range(B), 2), rng)


def test_config5_large_exponent(H):
    """BASELINE config 5: RSA-2048 with a 2048-bit exponent (2048 squarings + popcount(e) multiplies, one
    dependent chain per signature), batch 256 (about 50 GB of trace)."""
    chip = H.BigIntChip(64, 2048)
    o = Oracle(64, 32)
    rng = random.Random(0x68327273 + 5)
    B = 256
    e = rng.getrandbits(2048) | (1 << 2047)
    N = [rand_modulus(rng, 2048) for _ in range(B)]
    X = [rng.randrange(n) for n in N]
    res = chip.pow_mod_fixed_exp(chip.assign_integer(X), e, chip.assign_integer(N))
    assert res.trace.num_mul_mods == 2048 + bin(e).count("1")
    _check_pow_batch(H, chip, o, X, N, e, res, [0, B - 1], rng)


@pytest.mark.parametrize("w,L", [(64, 32), (32, 128)])
def test_assign_integer_range_check_sublimbs(H, w, L):
    """assign_integer range-checks every input limb (big_integer/chip.rs:71-76): 8 sub-limbs of w/8 bits."""
    chip = H.BigIntChip(w, w * L)
    rng = random.Random(w)
    vals = [rng.getrandbits(w * L) for _ in range(5)]
    a = chip.assign_integer(vals)
    sub = chip.range_check_sublimbs(a).cpu().numpy()
    sb = w // 8
    for i, v in enumerate(vals):
        for k in range(L):
            limb = (v >> (w * k)) & ((1 << w) - 1)
            assert [int(x) for x in sub[i, k]] == [(limb >> (sb * t)) & ((1 << sb) - 1) for t in range(8)]


@pytest.mark.parametrize("w,L", [(64, 32), (32, 128), (64, 16)])
def test_fresh_integer_family(H, w, L):
    """BigIntInstructions add / sub / add_mod / sub_mod / is_zero / is_equal_fresh / comparisons / is_in_field
    (reference big_integer/chip.rs:245-373, 452-528, 754-805, 908-1006; tests :1470-1660, 2395-2800):
    values, predicate bits and the whole op-trace vs the oracle, incl. the a+b == n and a == b corner cases."""
    from oracle_lib import FRESH_OPS, fresh_op
    chip = H.BigIntChip(w, w * L)
    o = Oracle(w, L)
    rng = random.Random(5 * w + L)
    bits = w * L
    n = rand_modulus(rng, bits)
    A = [rng.randrange(n) for _ in range(10)]
    B = [rng.randrange(n) for _ in range(10)]
    B[0] = n - A[0]          # a + b == n
    B[1] = A[1]              # a == b
    A[2] = 0                 # zero
    A[3], B[3] = n - 1, n - 1
    A[4], B[4] = 1, 0
    a_dev, b_dev, n_dev = chip.assign_integer(A), chip.assign_integer(B), chip.assign_integer([n])
    for name in FRESH_OPS:
        fn = getattr(chip, name)
        if name in ("add_mod", "sub_mod"):
            res = fn(a_dev, b_dev, n_dev)
        elif name == "is_zero":
            res = fn(a_dev)
        else:
            res = fn(a_dev, b_dev)
        torch.cuda.synchronize()
        st = res.status.cpu().tolist()
        fl = res.flag.cpu().tolist()
        vals = res.value.to_big_uint() if res.value is not None else None
        for i in range(10):
            rc, ov, of, ost = fresh_op(o, name, o.limbs(A[i]), o.limbs(B[i]), o.limbs(n))
            if rc != 0:
                assert st[i] != 0, (name, i)
                continue
            assert st[i] == 0, (name, i, st[i])
            assert np.array_equal(res.flatten(i), ost), (name, i)
            if vals is not None:
                assert vals[i] == o.to_int(ov), (name, i)
            if of >= 0:
                assert fl[i] == of, (name, i)
        if name == "add_mod":
            assert vals[5] == (A[5] + B[5]) % n and vals[0] == n          # un-reduced when a + b == n (as in the reference)
        if name == "sub_mod":
            assert vals[5] == (A[5] - B[5]) % n and vals[1] == n          # sub_mod(a, a, n) == n (as in the reference)
        if name == "is_less_than":
            assert fl == [int(A[i] < B[i]) for i in range(10)]
        if name == "is_greater_than_or_equal":
            assert fl == [int(A[i] >= B[i]) for i in range(10)]


def test_fresh_ops_with_one_shared_comparand(H):
    """is_in_field(a, n) / assert_in_field(a, n) / the comparisons with ONE `b` for the whole batch (the modulus): round 2's
    wrappers passed a batch-1 modulus as `b` while the kernel read b[elem] -- out of bounds.  Now H2R_F_SHARED_MODULUS on an
    op without `n` means a shared `b`; a batch that is neither 1 nor the operand's gets H2R_E_SHAPE."""
    from halo2_rsa_amd import _lib
    from oracle_lib import fresh_op
    chip = H.BigIntChip(64, 2048)
    o = Oracle(64, 32)
    rng = random.Random(606)
    n = rand_modulus(rng, 2048)
    A = [rng.randrange(n) for _ in range(37)] + [n, n + 1, n - 1]
    a_dev, n1 = chip.assign_integer(A), chip.assign_integer([n])
    n_rep = chip.assign_integer([n] * len(A))
    for name in ("is_in_field", "is_less_than", "is_greater_than_or_equal", "is_equal_fresh"):
        shared, rep = getattr(chip, name)(a_dev, n1), getattr(chip, name)(a_dev, n_rep)
        torch.cuda.synchronize()
        assert torch.equal(shared.flag, rep.flag) and torch.equal(shared.status, rep.status), name
        assert torch.equal(shared.trace, rep.trace), name
        for i in (0, 36, 37, 38, 39):
            rc, ov, of, ost = fresh_op(o, name, o.limbs(A[i]), o.limbs(n), o.limbs(n))
            assert rc == 0
            assert np.array_equal(shared.flatten(i), ost), (name, i)
            assert int(shared.flag[i]) == of, (name, i)
    res = chip.assert_in_field(a_dev, n1)
    st = res.status.cpu().tolist()
    assert st[:37] == [0] * 37 and st[37] == _lib.H2R_E_ASSERTION and st[38] == _lib.H2R_E_ASSERTION and st[39] == 0
    with pytest.raises(_lib.H2RError):
        chip.is_in_field(a_dev, chip.assign_integer([n, n]))


def test_constants_max_value_and_assertions(H):
    """instructions.rs:16-32, 197-254: assign_constant_fresh / assign_constant_muled / max_value and the assert_* family
    (predicate + main_gate.assert_one: a violated assertion -> status H2R_E_ASSERTION for that element only), exercised
    like the reference's tests do (big_integer/chip.rs:1470-1660: operands vs assigned constants)."""
    from halo2_rsa_amd import _lib
    chip = H.BigIntChip(64, 2048)
    rng = random.Random(41)
    n = rand_modulus(rng, 2048)
    vals = [rng.randrange(n) for _ in range(4)]
    a = chip.assign_integer(vals)
    c = chip.assign_constant_fresh(vals[2], batch=4)                      # the same constant in every element
    assert c.to_big_uint() == [vals[2]] * 4 and c.num_limbs() == 32
    assert chip.max_value(batch=2).to_big_uint() == [(1 << 2048) - 1] * 2   # chip.rs:138-154
    assert chip.max_value(3).to_big_uint() == [(1 << 192) - 1]
    with pytest.raises(AssertionError):
        chip.assign_constant(1 << 70, 1)                                  # does not fit (chip.rs:1266)
    r = chip.assert_equal_fresh(a, c)
    torch.cuda.synchronize()
    assert r.status.cpu().tolist() == [_lib.H2R_E_ASSERTION, _lib.H2R_E_ASSERTION, 0, _lib.H2R_E_ASSERTION]
    assert chip.assert_in_field(a, chip.assign_integer([n] * 4)).status.cpu().tolist() == [0, 0, 0, 0]
    assert chip.assert_less_than(a, c).status.cpu().tolist() == [0 if v < vals[2] else _lib.H2R_E_ASSERTION for v in vals]
    assert chip.assert_greater_than_or_equal(a, c).status.cpu().tolist() == [0 if v >= vals[2] else _lib.H2R_E_ASSERTION for v in vals]
    assert chip.assert_less_than_or_equal(a, c).status.cpu().tolist() == [0 if v <= vals[2] else _lib.H2R_E_ASSERTION for v in vals]
    assert chip.assert_greater_than(a, c).status.cpu().tolist() == [0 if v > vals[2] else _lib.H2R_E_ASSERTION for v in vals]
    z = chip.assign_constant_fresh(0, batch=4)
    assert chip.assert_zero(z).status.cpu().tolist() == [0] * 4 and chip.assert_zero(a).status.cpu().tolist() == [_lib.H2R_E_ASSERTION] * 4
    # mul(a, b) == the product assigned as a Muled constant ONLY when no column needs a carry: small operands
    x, y = 3, 5
    ax, ay = chip.assign_constant_fresh(x, 2), chip.assign_constant_fresh(y, 2)
    prod = chip.mul(ax, ay)
    cm = chip.assign_constant_muled(x * y, 32, 32, batch=2)
    st, _ = chip.assert_equal_muled(prod, cm)
    st2, _ = chip.assert_equal_muled(prod, chip.assign_constant_muled(x * y + 1, 32, 32, batch=2))
    torch.cuda.synchronize()
    assert st.cpu().tolist() == [0, 0] and st2.cpu().tolist() == [_lib.H2R_E_ASSERTION] * 2


def test_mul_reference_cases_on_gpu(H, golden):
    """BigIntChip::mul known answers of the reference (big_integer/chip.rs:2797-3107, incl. the 16-limb squaring
    with all 31 un-carried columns) through h2r_mul_batch."""
    from oracle_lib import mul_stream
    chip = H.BigIntChip(64, 2048)
    o = Oracle(64, 32)
    cases = golden["mul_cases"]
    A = [sum(int(x, 16) << (64 * i) for i, x in enumerate(c["a"])) for c in cases]
    B = [sum(int(x, 16) << (64 * i) for i, x in enumerate(c["b"])) for c in cases]
    res = chip.mul(chip.assign_integer(A), chip.assign_integer(B))
    torch.cuda.synchronize()
    for i, c in enumerate(cases):
        cols = res.columns(i)
        want = [int(x) for x in c["cols"]]
        assert cols[:len(want)] == want and all(v == 0 for v in cols[len(want):]), c["name"]
        ocols, ost = mul_stream(o, o.limbs(A[i]), o.limbs(B[i]))
        assert ocols == cols and np.array_equal(res.flatten(i), ost), c["name"]


@pytest.mark.parametrize("w,L", [(64, 32), (32, 128), (64, 16)])
def test_mul_is_equal_muled_refresh(H, w, L):
    """The reference's Muled-integer tests: ab == ba (test_muled_equal_circuit :1699), a bad twin (:1742), and
    refresh(ab) == refresh(ba) == a*b as Fresh limbs (test_refresh_circuit :1863-1890)."""
    from oracle_lib import is_equal_muled, mul_stream, refresh
    chip = H.BigIntChip(w, w * L)
    o = Oracle(w, L)
    rng = random.Random(9 * w + L)
    bits = w * L
    A = [rng.getrandbits(bits) for _ in range(6)]
    B = [rng.getrandbits(bits) for _ in range(6)]
    A[0] = B[0] = (1 << bits) - 1      # maximal limbs: every column at word_max
    A[1] = 0
    a_dev, b_dev = chip.assign_integer(A), chip.assign_integer(B)
    ab, ba = chip.mul(a_dev, b_dev), chip.mul(b_dev, a_dev)
    aa = chip.square(a_dev)
    eq, tr = chip.is_equal_muled(ab, ba)
    neq, trn = chip.is_equal_muled(ab, aa)           # a*b vs a*a: unequal unless a == b
    fresh, rstream, rstatus = chip.refresh(ab)
    torch.cuda.synchronize()
    assert eq.cpu().tolist() == [1] * 6
    assert neq.cpu().tolist() == [1 if A[i] * B[i] == A[i] * A[i] else 0 for i in range(6)]
    assert not rstatus.cpu().numpy().any()
    fr = fresh.to_big_uint()
    for i in range(6):
        assert fr[i] == A[i] * B[i]
        cols_ab, st_ab = mul_stream(o, o.limbs(A[i]), o.limbs(B[i]))
        cols_ba, _ = mul_stream(o, o.limbs(B[i]), o.limbs(A[i]))
        cols_aa, st_aa = mul_stream(o, o.limbs(A[i]), o.limbs(A[i]))
        assert ab.columns(i) == cols_ab and np.array_equal(ab.flatten(i), st_ab)
        assert np.array_equal(aa.flatten(i), st_aa)
        e, est = is_equal_muled(o, cols_ab, cols_ba)
        assert e == 1 and np.array_equal(chip.flatten_is_equal_muled(tr, i), est), i
        e, est = is_equal_muled(o, cols_ab, cols_aa)
        assert e == int(neq[i].item()) and np.array_equal(chip.flatten_is_equal_muled(trn, i), est), i
        rc, rl, rst = refresh(o, cols_ab)
        assert rc == 0 and np.array_equal(rstream[i].cpu().numpy(), rst), i


def test_dist_exports_one_rank(H):
    """h2r_dist_* (RCCL behind the C ABI) with a one-rank communicator on this box's GPU, through the Python mirror bench.py uses
    for N > 1 (halo2_rsa_amd.dist.H2RDist): parameter broadcast, all-gather of the per-signature results, MAX, barrier; and
    the shard ranges against halo2_rsa_amd.dist.shard_range."""
    import ctypes
    from halo2_rsa_amd import _lib
    from halo2_rsa_amd.dist import H2RDist, shard_range
    chip = H.BigIntChip(64, 2048)
    lo, hi = ctypes.c_uint64(), ctypes.c_uint64()
    for total in (0, 1, 7, 1024, 65536, 65537):
        for world in (1, 2, 3, 8):
            for r in range(world):
                assert _lib.lib().h2r_dist_shard_range(total, r, world, ctypes.byref(lo), ctypes.byref(hi)) == 0
                assert (lo.value, hi.value) == shard_range(total, r, world)
    assert _lib.lib().h2r_dist_shard_range(10, 3, 3, ctypes.byref(lo), ctypes.byref(hi)) == _lib.H2R_E_SHAPE
    d = H2RDist(chip, 0, 1, 0)
    assert d.broadcast_ints([65537, 2048, 4, 20, 5]) == [65537, 2048, 4, 20, 5]
    assert d.max_over_ranks(0.75) == 0.75
    d.barrier()
    rng = random.Random(12)
    N = [rand_modulus(rng, 2048) for _ in range(5)]
    X = [rng.randrange(n) for n in N]
    res = chip.pow_mod_fixed_exp(chip.assign_integer(X), 65537, chip.assign_integer(N), want_trace=False)
    g = d.gather_to_rank0(res.value.limbs_dev)
    assert torch.equal(g, res.value.limbs_dev)
    assert H.AssignedInteger(g, 64).to_big_uint() == [pow(x, 65537, n) for x, n in zip(X, N)]
    d.finalize()


def _u256_tensor(cols_list, ncols):
    arr = np.zeros((len(cols_list), ncols, 4), dtype=np.uint64)
    for e, cols in enumerate(cols_list):
        for c, v in enumerate(cols):
            for k in range(4):
                arr[e, c, k] = (v >> (64 * k)) & (2 ** 64 - 1)
    return torch.from_numpy(arr.view(np.int64)).cuda()


@pytest.mark.parametrize("w,L,shapes", [(64, 32, [(32, 32), (7, 32), (32, 1), (5, 12), (1, 1), (16, 17)]),
                                        (32, 128, [(128, 128), (3, 128), (100, 9)]), (64, 8, [(8, 8), (2, 5)])])
def test_general_operand_shapes_mul_refresh_is_equal_muled(H, w, L, shapes):
    """The reference's mul takes d0 != d1 (big_integer/chip.rs:395-397), refresh any RefreshAux::new(w, n_l, n_r) (mod.rs:428,
    chip.rs:178-181), is_equal_muled n_l != n_r with word_max from min(n_l, n_r) (chip.rs:822-842): h2r_mul_batch_ex,
    h2r_refresh_batch_ex (parallel fixed-point carries), h2r_is_equal_muled_batch_ex against oracle/pyref.py value for value --
    columns, Fresh limbs (== the product as an integer), every streamed intermediate, equal and unequal pairs, a_b as a field
    element, and a limb that does not fit RefreshAux (status)."""
    import os
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
    import pyref as R
    from halo2_rsa_amd import _lib
    chip = H.BigIntChip(w, w * L)
    p = R.Params(w, L)
    pf = R.Params(w, L, field_modulus=R.FIELD_MODULI["bn254_fr"])
    rng = random.Random(w * L + len(shapes))
    np_dt = np.uint64 if w == 64 else np.uint32
    for (d0, d1) in shapes:
        batch = 4
        A = [[rng.getrandbits(w) for _ in range(d0)] for _ in range(batch)]
        Bs = [[rng.getrandbits(w) for _ in range(d1)] for _ in range(batch)]
        A[1] = [(1 << w) - 1] * d0                       # all-ones operands: the largest columns, long carry ripples in refresh
        Bs[1] = [(1 << w) - 1] * d1
        A[2] = [0] * d0
        to_dev = lambda rows: H.AssignedInteger(torch.from_numpy(np.array(rows, dtype=np_dt).view(np.int64 if w == 64 else np.int32)).cuda(), w)
        m = chip.mul_ex(to_dev(A), to_dev(Bs))
        torch.cuda.synchronize()
        ncols = d0 + d1 - 1
        cols_host = m.cols.cpu().numpy().view(np.uint64)
        ref_cols = []
        for e in range(batch):
            st = R.Stream()
            cols = R.mul_columns(A[e], Bs[e], st, p.WB)
            ref_cols.append(cols)
            got = [sum(int(cols_host[e, c, k]) << (64 * k) for k in range(4)) for c in range(ncols)]
            assert got == cols, (d0, d1, e)
            assert bytes(chip.mul_ex_flatten(m, e)) == st.bytes(), (d0, d1, e)
        # refresh with RefreshAux::new(w, d0, d1)
        fresh, streams, status = chip.refresh_ex(m.cols, d0, d1)
        torch.cuda.synchronize()
        assert not status.cpu().numpy().any()
        fl = fresh.to_big_uint()
        sh = streams.cpu().numpy()
        for e in range(batch):
            st = R.Stream()
            want = R.refresh(p, ref_cols[e], st, d0, d1)
            assert fl[e] == R.from_limbs(want, w) == R.from_limbs(A[e], w) * R.from_limbs(Bs[e], w), (d0, d1, e)
            assert bytes(sh[e]) == st.bytes(), (d0, d1, e)
        # is_equal_muled(n_l = d0, n_r = d1): a * b against itself, and against a copy with one column changed
        other = [list(c) for c in ref_cols]
        other[0][ncols // 2] += 3
        other[3][ncols - 1] = max(0, other[3][ncols - 1] - 1) if other[3][ncols - 1] else 1
        a_dev, b_dev = _u256_tensor(ref_cols, 2 * L), _u256_tensor(other, 2 * L)
        for flags, pp in ((0, p), (1, pf)):
            for (x_dev, y_dev, ys) in ((a_dev, a_dev, ref_cols), (a_dev, b_dev, other), (b_dev, a_dev, None)):
                out, eq = chip.is_equal_muled_ex(x_dev, y_dev, d0, d1, flags)
                torch.cuda.synchronize()
                oh, eqh = out.cpu().numpy(), eq.cpu().tolist()
                for e in range(batch):
                    xa = ref_cols[e] if x_dev is a_dev else other[e]
                    yb = (ref_cols[e] if y_dev is a_dev else other[e])
                    st = R.Stream()
                    want_eq = R.is_equal_muled(pp, xa, yb, st, d0, d1)
                    assert eqh[e] == want_eq, (d0, d1, e, flags)
                    assert bytes(oh[e]) == st.bytes(), (d0, d1, e, flags)
    # a Muled limb too large for RefreshAux::new(w, 2, 2): assert_zero(limb) fails (chip.rs:213) -> status, other elements fine
    big = [[1 << (2 * w + 9), 5, 6], [7, 8, 9]]
    fresh, streams, status = chip.refresh_ex(_u256_tensor(big, 3), 2, 2)
    torch.cuda.synchronize()
    assert status.cpu().tolist() == [_lib.H2R_E_NOT_REDUCED, 0]
    assert fresh.to_big_uint()[1] == 7 + (8 << w) + (9 << (2 * w))
    with pytest.raises(_lib.H2RError):
        chip.refresh_ex(_u256_tensor(big, 3), L + 1, 1)          # operands longer than the chip's num_limbs


@pytest.mark.parametrize("B,NL,EB,bits,form", [(48, 5, 13, 2048, (2, 1)), (640, 1, 5, 2048, (2, 1)), (6, 10, 60, 2048, (2, 1)),   # (600-bit exponents: walked as two segments of bits)
                                               (1408, 1, 5, 1024, (3, 2)),    # [r6] RSA-1024, two-queue form: one-wave chains walking per-element exponents
                                               (1024, 1, 5, 2048, (3, 2))])   # RSA-2048, two-queue form
def test_pipelined_variable_exponent_calls(H, B, NL, EB, bits, form):
    """h2r_pipeline_modpow_public_key_var (RSAPubE::Var, src/chip.rs:108-110): three pipelined calls with per-element 5-limb x 13-bit
    exponents over the pipeline's buffer sets leave byte for byte what the stream-ordered export writes (trace incl. e bits and selected
    limbs, in-field witness, results, status), and the results are pow(x, e, n)."""
    # (640 per call: issued as one-launch steps -- the chain role runs the variable-exponent walk inside step_kernel)
    chip = H.BigIntChip(64, bits)
    rng = random.Random(808 + B)
    pl = chip.pow_var_layout(NL, EB)
    ies = chip.in_field_layout()[0]
    mk = lambda nbytes: torch.zeros(nbytes, dtype=torch.uint8, device="cuda")
    sets = [dict(trace=mk(B * pl.elem_stride), inf=mk(B * ies), ws=mk(chip.workspace_bytes(B, pl.num_mul_mods)),
                 out=torch.zeros((B, bits // 64), dtype=torch.int64, device="cuda"), status=mk(B)) for _ in range(form[0])]
    pipe = H.Pipeline(chip, depth=form[0], side_streams=form[1])   # (as many buffer sets as the pipeline is deep)
    calls, snaps = [], []
    for k in range(3):
        N = [rand_modulus(rng, bits) for _ in range(min(B, 64))]
        N = [N[i % len(N)] for i in range(B)]
        X = [rng.randrange(n) for n in N]
        E = [[rng.getrandbits(EB) for _ in range(NL)] for _ in range(B)]
        e_dev = H.AssignedInteger(torch.tensor(E, dtype=torch.int64, device="cuda"), 64)
        calls.append((N, X, E, chip.assign_integer(N), chip.assign_integer(X), e_dev))
        s = sets[k % form[0]]
        if k >= form[0]:
            snaps.append((s["trace"].clone(), s["inf"].clone(), s["out"].clone(), s["status"].clone()))
        pipe.modpow_public_key_var(calls[k][4], e_dev, EB, calls[k][3], s["trace"], s["ws"], s["out"], s["status"], in_field_buf=s["inf"])
    pipe.join()
    for k in range(max(0, 3 - form[0]), 3):
        s = sets[k % form[0]]
        snaps.append((s["trace"].clone(), s["inf"].clone(), s["out"].clone(), s["status"].clone()))
    torch.cuda.synchronize()
    for k in range(3):
        N, X, E, n_dev, x_dev, e_dev = calls[k]
        trace, inf, out, status = snaps[k]
        ref = chip.pow_mod(x_dev, e_dev, n_dev, EB, check_in_field=True)
        torch.cuda.synchronize()
        assert not status.cpu().numpy().any()
        # (the stream-ordered export's buffers come from torch.empty: compare the flat streams -- every witness byte -- not the padding)
        assert torch.equal(ref.trace.emit_stream(), H.Trace(chip, trace, B, pl).emit_stream()), k
        assert torch.equal(ref.value.limbs_dev, out), k
        for i in (0, B // 2, B - 1):
            assert np.array_equal(ref.in_field.flatten(i), H.big_integer.InFieldTrace(chip, inf, B, ies, chip.in_field_layout()[1]).flatten(i)), (k, i)
        got = H.AssignedInteger(out, 64).to_big_uint()
        assert all(got[i] == pow(X[i], sum(v << (EB * j) for j, v in enumerate(E[i])), N[i]) for i in range(B)), k
    pipe.close()


@pytest.mark.gpu
@pytest.mark.parametrize("w,L,field", [(64, 32, "bn254_fr"), (32, 16, "pasta_fq"), (64, 16, "bn254_fq")])
def test_fresh_family_advice_rows(H, w, L, field):
    """h2r_fresh_op_emit_advice: the rows of EVERY op of the Fresh-integer family (and of the asserting variants) -- every cell
    -- equal the image built in Python from the ORACLE's stream of the op (tests/advice_ref.fresh_image), incl. the a == b,
    a + b == n and zero corner cases and main_gate.is_zero's inverse witnesses; and the GPU's cells satisfy the main-gate
    equation with the fixed rows the C ABI reports, with every range row's cells in the lookup table."""
    import ctypes
    import os
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
    import pyref as R
    import advice_ref as AR
    from halo2_rsa_amd import _lib
    from halo2_rsa_amd._lib import lib
    from oracle_lib import FRESH_OPS, fresh_op
    chip = H.BigIntChip(w, w * L, field=field)
    o = Oracle(w, L)
    P = R.FIELD_MODULI[field]
    rng = random.Random(11 * w + L)
    n = rand_modulus(rng, w * L)
    A = [rng.randrange(n) for _ in range(6)]
    B = [rng.randrange(n) for _ in range(6)]
    B[0] = n - A[0]
    B[1] = A[1]
    A[2] = 0
    A[3], B[3] = 1, 0
    if A[4] > B[4]:
        A[4], B[4] = B[4], A[4]          # a < b: the asserting variants of the comparisons hold for this element
    a_dev, b_dev, n_dev = chip.assign_integer(A), chip.assign_integer(B), chip.assign_integer([n])
    cfg = AR.LookupConfig(AR.range_lens(w, L, rsa=(w == 64)))
    la = H.LookupArgument(chip, rsa_chip=(w == 64))
    table = set(cfg.table())
    fixed = {}
    for name in FRESH_OPS:
        fn = getattr(chip, name)
        res = fn(a_dev, b_dev, n_dev) if name in ("add_mod", "sub_mod") else (fn(a_dev) if name == "is_zero" else fn(a_dev, b_dev))
        variants = [False] if name in ("add", "sub", "add_mod", "sub_mod") else [False, True]
        for assert_one in variants:
            img = res.emit_advice(assert_one=assert_one)
            torch.cuda.synchronize()
            kinds = chip.fresh_op_row_kinds(res.op, assert_one)
            host = img.cpu().numpy()
            st = res.status.cpu().tolist()
            for i in range(6):
                rc, ov, of, ost = fresh_op(o, name, o.limbs(A[i]), o.limbs(B[i]), o.limbs(n))
                if rc != 0:
                    assert st[i] != 0
                    continue
                im = AR.fresh_image(o.p, name, o.limbs(A[i]), None if name == "is_zero" else o.limbs(B[i]),
                                    o.limbs(n) if name in ("add_mod", "sub_mod") else None, ost, P, assert_one=assert_one)
                assert im.kinds == kinds.tolist()
                want = AR.image_bytes(im)
                got = host[i].reshape(len(kinds), 160)
                if not np.array_equal(got, want):
                    bad = np.argwhere(got != want)[0]
                    pytest.fail("%s w=%d L=%d elem %d: row %d (kind %d) cell %d differs" % (name, w, L, i, int(bad[0]), int(kinds[int(bad[0])]), int(bad[1]) // 32))
                if i in (1, 4) and not assert_one:   # gate equation + lookup membership on the GPU's own cells
                    rows = len(kinds)
                    cells = [[int.from_bytes(got[r, 32 * c:32 * c + 32].tobytes(), "little") for c in range(5)] for r in range(rows)]
                    for r in range(rows):
                        k = int(kinds[r])
                        if k not in fixed:
                            fr = _lib.H2RFixedRow()
                            assert lib().h2r_advice_fixed_row(chip._ctx, ctypes.byref(la.cfg), k, ctypes.byref(fr)) == 0
                            fixed[k] = fr.as_dict()
                        f = fixed[k]
                        assert AR.gate_residual(cells[r], cells[r + 1][4] if r + 1 < rows else 0, f, P) == 0, (name, r, k)
                        if f["tag_composition"]:
                            assert all((f["tag_composition"], cells[r][c]) in table for c in range(4)), (name, r)


@pytest.mark.gpu
def test_assert_in_field_advice_rows_of_a_modpow_call(H):
    """The in_field_trace of modpow_public_key as advice rows (InFieldTrace.emit_advice -> h2r_fresh_op_emit_advice with
    H2R_ADVICE_ASSERT_ONE, shared modulus): equal to advice_ref.in_field_image of the oracle's assert_in_field stream; 1,532 rows for
    RSA-2048; the last row ([lt], assert_one) is the only unsatisfied one for x >= n."""
    import os
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
    import pyref as R
    import advice_ref as AR
    chip = H.BigIntChip(64, 2048)
    o = Oracle(64, 32)
    P = R.FIELD_MODULI["bn254_fr"]
    rng = random.Random(2048)
    n = rand_modulus(rng, 2048)
    X = [rng.randrange(n) for _ in range(5)] + [n + 5]
    x_dev, n_dev = chip.assign_integer(X), chip.assign_integer([n])
    res = chip.pow_mod_fixed_exp(x_dev, 65537, n_dev, check_in_field=True)     # = RSAChip::modpow_public_key (src/chip.rs:99-114)
    img = res.in_field.emit_advice(x_dev, n_dev)
    torch.cuda.synchronize()
    assert res.status.cpu().tolist()[:5] == [0] * 5 and int(res.status[5]) != 0
    host = img.cpu().numpy()
    assert img.shape[1] == 1532 * 160
    for i in range(6):
        rc, lt, st = o.assert_in_field(o.limbs(X[i]), o.limbs(n))
        im = AR.in_field_image(o.p, o.limbs(X[i]), o.limbs(n), st, P)
        assert np.array_equal(host[i].reshape(1532, 160), AR.image_bytes(im)), i
        assert int.from_bytes(host[i].reshape(1532, 160)[-1, :32].tobytes(), "little") == (1 if X[i] < n else 0)


@pytest.mark.gpu
def test_verify_element_advice_image(H, golden):
    """h2r_verify_emit_advice: one whole verify_pkcs1v15_signature element as advice rows -- [is_eq = assign_constant(1)]
    [assert_in_field] [pow_mod_fixed_exp] [encoded-message check] -- for the reference's KATs (valid, valid, bad) and random
    signatures: the in-field and EM sections equal the Python restatements of the ORACLE's streams, the pow section is the image
    h2r_pow_trace_emit_advice writes (checked cell by cell in test_advice_image_*), the row kinds are the sections' kinds, and
    the GPU's EM cells satisfy the main gate with the fixed rows of the C ABI (incl. the 4-bit range rows' table membership)."""
    import ctypes
    import os
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
    import pyref as R
    import advice_ref as AR
    from halo2_rsa_amd import _lib
    from halo2_rsa_amd._lib import lib
    rsa = H.RSAChip(2048, 5)
    chip = rsa.bigint_chip()
    P = R.FIELD_MODULI["bn254_fr"]
    kats = golden["rsa_kats"]
    rng = random.Random(77)
    ns = [int(k["n"]) for k in kats] + [rand_modulus(rng, 2048) for _ in range(2)]
    sigs = [int(k["sig"]) for k in kats] + [rng.randrange(n) for n in ns[3:]]
    hashed = [int(k["hashed"]) for k in kats] + [rng.getrandbits(256) for _ in range(2)]
    sigs[4] = ns[4] + 1                                          # not in field: skipped
    pk = rsa.assign_public_key(H.RSAPublicKey(H.UnassignedInteger.from_ints(ns, 32, 64), H.Fix(65537)))
    sg = rsa.assign_signature(H.RSASignature(H.UnassignedInteger.from_ints(sigs, 32, 64)))
    res = rsa.verify_pkcs1v15_signature(pk, hashed, sg)
    total, sec = res.advice_sections()
    assert sec == [1, 1532, 2 + 19 * 3974, 178] and total == sum(sec)
    img = res.emit_advice()
    torch.cuda.synchronize()
    assert res.status.cpu().tolist() == [0, 0, 0, 0, H.H2R_E_NOT_IN_FIELD]
    host = img.cpu().numpy().reshape(5, total, 160)
    kinds = res.row_kinds()
    o = Oracle(64, 32)
    # the pow section straight from the export that is checked cell by cell elsewhere
    pow_img = torch.empty((5, sec[2] * 160), dtype=torch.uint8, device=img.device)
    sig_d, n_d, hashed_d = res.inputs
    assert lib().h2r_pow_trace_emit_advice(chip._ctx, ctypes.byref(res.layout.pow), n_d.data_ptr(), chip._flags(n_d, 5), res.trace.data_ptr(),
                                           res.layout.elem_stride, res.workspace.data_ptr(), 5, res.status.data_ptr(), pow_img.data_ptr(),
                                           pow_img.shape[1], chip._stream()) == 0
    torch.cuda.synchronize()
    pow_host = pow_img.cpu().numpy().reshape(5, sec[2], 160)
    la = H.LookupArgument(chip, rsa_chip=True)
    cfg = AR.LookupConfig(AR.range_lens(64, 32, rsa=True))
    table = set(cfg.table())
    for i in (0, 1, 2, 3):
        assert int.from_bytes(host[i, 0, :32].tobytes(), "little") == 1 and not host[i, 0, 32:].any()
        rc, lt, s_if = o.assert_in_field(o.limbs(sigs[i]), o.limbs(ns[i]))
        im_if = AR.in_field_image(o.p, o.limbs(sigs[i]), o.limbs(ns[i]), s_if, P)
        assert np.array_equal(host[i, 1:1 + sec[1]], AR.image_bytes(im_if)), ("in_field", i)
        assert np.array_equal(host[i, 1 + sec[1]:1 + sec[1] + sec[2]], pow_host[i]), ("pow", i)
        powed = o.limbs(pow(sigs[i], 65537, ns[i]))
        rc, valid, s_em = o.pkcs1v15_em_check(powed, o.limbs(hashed[i], 4))
        im_em, is_eq = AR.em_image(o.p, powed, o.limbs(hashed[i], 4), s_em, P)
        got = host[i, total - sec[3]:]
        want = AR.image_bytes(im_em)
        if not np.array_equal(got, want):
            bad = np.argwhere(got != want)[0]
            pytest.fail("EM elem %d: row %d (kind %d) cell %d differs" % (i, int(bad[0]), im_em.kinds[int(bad[0])], int(bad[1]) // 32))
        assert is_eq == int(res.is_valid[i]) == (1 if i < 2 else 0)
        assert kinds[0] == AR.ROW_CONST1 and kinds[1:1 + sec[1]].tolist() == im_if.kinds and kinds[total - sec[3]:].tolist() == im_em.kinds
        if i == 0:
            assert kinds[1 + sec[1]] == AR.ROW_CONST1 and kinds[2 + sec[1]] == AR.ROW_CONST0
            cells = [[int.from_bytes(got[r, 32 * c:32 * c + 32].tobytes(), "little") for c in range(5)] for r in range(sec[3])]
            for r in range(sec[3]):
                fr = _lib.H2RFixedRow()
                assert lib().h2r_advice_fixed_row(chip._ctx, ctypes.byref(la.cfg), im_em.kinds[r], ctypes.byref(fr)) == 0
                f = fr.as_dict()
                ref = AR.fixed_row(im_em.kinds[r], 64, 32, o.p.carry_bits, o.p.carry_sub_bits, o.p.carry_nsub, cfg)
                assert {nm: v % P for nm, v in ref.items() if nm in AR.FIXED_NAMES} == {nm: f[nm] for nm in AR.FIXED_NAMES}
                assert ref["tag_composition"] == f["tag_composition"]
                assert AR.gate_residual(cells[r], cells[r + 1][4] if r + 1 < sec[3] else 0, f, P) == 0, (r, im_em.kinds[r])
                if f["tag_composition"]:
                    assert all((f["tag_composition"], cells[r][c]) in table for c in range(4)), r


@pytest.mark.gpu
@pytest.mark.parametrize("w,L,dense", [(64, 32, True), (64, 32, False), (64, 16, True), (64, 48, True), (32, 128, True)])
def test_long_exponent_walked_as_segments(H, w, L, dense):
    """A long exponent on a latency-bound batch is walked as SEGMENTS of its bits (chain kernel of a segment, then its record kernel
    next to the following segment's chains; the running (squared, acc) pair crosses launches in the workspace): the plain export and
    three pipelined calls of a 700-bit exponent leave exactly the oracle's stream for every element (chip.rs:710-742), results =
    pow(x, e, n), every record passes the in-place audit, and an element with x >= n keeps its status through the later segments.
    dense / sparse exponents and 64- / 32-digit chains take the three chain builds (two chains side by side, deep, throughput)."""
    chip = H.BigIntChip(w, w * L)
    o = Oracle(w, L)
    rng = random.Random(4242 + L + dense)
    bits = w * L
    e = (rng.getrandbits(700) | (1 << 699)) if dense else ((1 << 699) | (1 << 350) | (1 << 33) | 1)
    B = 5
    N = [rand_modulus(rng, bits, odd=(i != 1)) for i in range(B)]
    X = [rng.randrange(n) for n in N]
    res = chip.pow_mod_fixed_exp(chip.assign_integer(X), e, chip.assign_integer(N))
    assert res.trace.num_mul_mods == 700 + bin(e).count("1")
    _check_pow_batch(H, chip, o, X, N, e, res, list(range(B)), rng)
    ref_stream = res.trace.emit_stream().clone()
    # one key, many elements (H2R_F_SHARED_MODULUS: for the 96-digit chains the Barrett constants come from recip_kernel, once per segment launch)
    Xs = [x % N[0] for x in X]
    shared = chip.pow_mod_fixed_exp(chip.assign_integer(Xs), e, chip.assign_integer(N[:1]))
    torch.cuda.synchronize()
    assert not shared.status.cpu().numpy().any()
    assert shared.value.to_big_uint() == [pow(x, e, N[0]) for x in Xs]
    bad, _first = shared.audit()
    torch.cuda.synchronize()
    assert not bad.cpu().numpy().any()
    rc, _oo, ost = o.pow_mod_fixed_exp(o.limbs(Xs[B - 1]), o.limbs(N[0]), e)
    assert rc == 0 and np.array_equal(ost, shared.trace.flatten(B - 1))
    del shared
    # pipelined, with the in-field check: element 2 of the second call has x >= n
    pl = chip.pow_fixed_layout(e)
    ies = chip.in_field_layout()[0]
    mk = lambda nbytes: torch.zeros(nbytes, dtype=torch.uint8, device="cuda")
    sets = [dict(trace=mk(B * pl.elem_stride), inf=mk(B * ies), ws=mk(chip.workspace_bytes(B, pl.num_mul_mods)),
                 out=torch.zeros((B, L), dtype=chip.torch_dtype, device="cuda"), status=mk(B)) for _ in range(2)]
    pipe = chip.pipeline()
    X2 = list(X)
    X2[2] = N[2] + 5 if N[2] + 5 < (1 << bits) else N[2]
    x_dev, x2_dev, n_dev = chip.assign_integer(X), chip.assign_integer(X2), chip.assign_integer(N)
    snaps = []
    for k in range(3):
        s = sets[k % 2]
        if k >= 2:
            snaps.append((s["trace"].clone(), s["out"].clone(), s["status"].clone()))
        pipe.modpow_public_key(x2_dev if k == 1 else x_dev, e, n_dev, s["trace"], s["ws"], s["out"], s["status"], s["inf"])
    pipe.join()
    for k in (1, 2):
        s = sets[k % 2]
        snaps.append((s["trace"].clone(), s["out"].clone(), s["status"].clone()))
    torch.cuda.synchronize()
    from halo2_rsa_amd import _lib
    for k in range(3):
        trace, out, status = snaps[k]
        st = status.cpu().tolist()
        got = H.AssignedInteger(out, w).to_big_uint()
        streams = H.Trace(chip, trace, B, pl).emit_stream()
        for i in range(B):
            if k == 1 and i == 2:
                assert st[i] == _lib.H2R_E_NOT_IN_FIELD
                continue
            assert st[i] == 0
            assert got[i] == pow(X[i], e, N[i]), (k, i)
            assert torch.equal(streams[i], ref_stream[i]), (k, i)
    pipe.close()
