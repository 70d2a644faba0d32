"""The caller of the accelerated path: RSASignatureVerifier::verify_pkcs1v15_signature (reference src/lib.rs:183-246) --
SHA-256 of the signed message, the reversed digest packed into the four hashed-message limbs (:210-239), then
RSAChip::verify_pkcs1v15_signature.  CPU tests pin the two restatements (oracle/h2r_oracle.c, oracle/pyref.py) against
tests/golden/sha256_kat.json (FIPS 180-4 example digests + the reference's own SHA-256("hello world") constant,
src/chip.rs:713) and against each other; GPU tests compare sha256_kernel / h2r_signature_verifier_batch with them."""
import ctypes
import hashlib
import json
import os
import random
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import pyref as R  # noqa: E402
import oracle_lib as OL  # noqa: E402
from oracle_lib import Oracle  # noqa: E402


@pytest.fixture(scope="module")
def sha_kat():
    with open(os.path.join(ROOT, "tests", "golden", "sha256_kat.json")) as f:
        return json.load(f)["vectors"]


# message lengths around every padding boundary (55/56: one vs two tail blocks; 64 k + ...), empty included
EDGE_LENS = [0, 1, 2, 3, 4, 5, 31, 32, 54, 55, 56, 57, 62, 63, 64, 65, 66, 67, 118, 119, 120, 121, 127, 128, 129, 183, 184, 191, 192, 193,
             255, 256, 257, 1000, 4099]


def test_sha256_restatements_against_golden(sha_kat):
    """Both oracles reproduce every known answer: digest, hashed-message limbs, the composition stream."""
    for v in sha_kat:
        msg = (v["msg"] * v["repeat"]).encode()
        want = bytes.fromhex(v["digest"])
        assert OL.sha256(msg) == want and R.sha256(msg) == want, v["msg"][:12]
        limbs, st = OL.hashed_msg(want)
        pst = R.Stream()
        assert [int(x) for x in limbs] == R.hashed_msg(want, pst) == [int(x) for x in v["hashed_limbs"]]
        assert st.tobytes() == pst.bytes() and hashlib.sha256(st.tobytes()).hexdigest() == v["stream_sha256"]
        assert len(st) == OL.lib().h2ro_hashed_msg_stream_bytes() == 288


def test_sha256_restatements_agree_on_ragged_lengths():
    """C restatement == Python restatement == hashlib on every edge length (the standard's padding cases)."""
    rng = random.Random(0x68327273)
    for n in EDGE_LENS:
        msg = bytes(rng.getrandbits(8) for _ in range(n))
        d = OL.sha256(msg)
        assert d == hashlib.sha256(msg).digest(), n
        if n <= 300:
            assert d == R.sha256(msg), n


def test_hashed_msg_is_the_reference_operand(golden, sha_kat):
    """src/chip.rs:713: the reference's hashed_msg constant is SHA-256("hello world") read big-endian; limb i of the packing is
    what RSAChip::verify_pkcs1v15_signature compares with powed limb i (src/chip.rs:141-144)."""
    d = OL.sha256(b"hello world")
    limbs, _ = OL.hashed_msg(d)
    for k in golden["rsa_kats"]:
        assert sum(int(v) << (64 * i) for i, v in enumerate(limbs)) == int(k["hashed"])


def test_hashed_msg_row_program_on_a_host_context():
    """h2r_hashed_msg_row_kinds (host-only: the symbolic walk of src/lib.rs:225-239) equals the kinds of the Python restatement;
    the device exports refuse a context without a device; a 32-bit-limb context is refused (RSAChip::LIMB_WIDTH = 64)."""
    import advice_ref as AR
    from test_cabi_host import host_ctx
    from halo2_rsa_amd import _lib
    from halo2_rsa_amd._lib import lib
    ctx = host_ctx(64, 32)
    rows = int(lib().h2r_hashed_msg_advice_rows(ctx))
    assert rows == 68
    kinds = np.zeros(rows, dtype=np.uint8)
    assert lib().h2r_hashed_msg_row_kinds(ctx, kinds.ctypes.data) == 0
    _, st = OL.hashed_msg(OL.sha256(b"hello world"))
    im, _ = AR.hashed_msg_image(st, R.FIELD_MODULI["bn254_fr"])
    assert im.kinds == kinds.tolist()
    buf = np.zeros(512, dtype=np.uint8)
    assert lib().h2r_sha256_hashed_msg_batch(ctx, buf.ctypes.data, None, 4, 1, None, None, None, 0, None) == _lib.H2R_E_UNSUPPORTED
    assert lib().h2r_hashed_msg_emit_advice(ctx, buf.ctypes.data, 0, 1, None, buf.ctypes.data, 68 * 160, None) == _lib.H2R_E_UNSUPPORTED
    assert lib().h2r_hashed_msg_advice_rows(host_ctx(32, 64)) == 0
    lib().h2r_ctx_destroy(ctx)


# ---------------------------------------------------------------------------------------------------------------- GPU
@pytest.fixture(scope="module")
def H():
    torch = pytest.importorskip("torch")
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    import halo2_rsa_amd as H
    return H


def _pack_with_gaps(msgs, rng, align=None):
    """Messages at arbitrary (or `align`-aligned) byte offsets of one buffer with junk between them: offsets [batch + 1] cannot
    express gaps, so the junk is part of no message -- the buffer is [junk0][m0][m1]...; `lead` junk bytes shift every start."""
    lead = rng.randrange(1, 9) if align is None else align
    buf = bytes(rng.getrandbits(8) for _ in range(lead)) + b"".join(msgs)
    off = np.zeros(len(msgs) + 1, dtype=np.int64)
    off[0] = lead
    np.cumsum([len(m) for m in msgs], out=off[1:])
    off[1:] += lead
    return np.frombuffer(buf, dtype=np.uint8).copy(), off


@pytest.mark.gpu
def test_sha256_kernel_known_answers_and_edge_lengths(H, sha_kat):
    """sha256_kernel on the FIPS vectors (incl. the empty message and one million 'a'), on every padding edge length, at
    unaligned and aligned message starts: digests, limbs and the 288-byte stream equal the C oracle's, bit for bit."""
    import torch
    chip = H.BigIntChip(64, 2048)
    rng = random.Random(5)
    msgs = [(v["msg"] * v["repeat"]).encode() for v in sha_kat]
    msgs += [bytes(rng.getrandbits(8) for _ in range(n)) for n in EDGE_LENS]
    msgs += [b"", b"", bytes(64), bytes(128)]
    for align in (None, 4, 16):
        buf, off = _pack_with_gaps(msgs, rng, align)
        dev = torch.device("cuda", 0)
        digest, hashed, hm = H.sha256_hashed_msg(chip, (torch.from_numpy(buf).to(dev), torch.from_numpy(off).to(dev)))
        torch.cuda.synchronize()
        digest, hashed, hm = digest.cpu().numpy(), hashed.cpu().numpy().view(np.uint64), hm.cpu().numpy()
        for i, m in enumerate(msgs):
            want = OL.sha256(m)
            assert digest[i].tobytes() == want, (align, i, len(m))
            limbs, st = OL.hashed_msg(want)
            assert np.array_equal(hashed[i], limbs) and np.array_equal(hm[i], st), (align, i)
        for i, v in enumerate(sha_kat):
            assert digest[i].tobytes().hex() == v["digest"]


@pytest.mark.gpu
def test_sha256_fixed_length_form_and_argument_checks(H):
    """msg_off = NULL: every message has fixed_len bytes (the reference's own tests sign 128-byte messages, src/lib.rs:339-343);
    nullable outputs; misaligned outputs and a short stride are refused before any launch; batch 0 is a no-op."""
    import torch
    from halo2_rsa_amd._lib import lib
    chip = H.BigIntChip(64, 2048)
    rng = random.Random(6)
    B, n = 300, 128
    raw = np.frombuffer(bytes(rng.getrandbits(8) for _ in range(B * n)), dtype=np.uint8).copy()
    buf = torch.from_numpy(raw).cuda()
    digest = torch.zeros((B, 32), dtype=torch.uint8, device="cuda")
    hashed = torch.zeros((B, 4), dtype=torch.int64, device="cuda")
    st = chip._stream()
    assert lib().h2r_sha256_hashed_msg_batch(chip._ctx, buf.data_ptr(), None, n, B, digest.data_ptr(), None, None, 0, st) == 0
    assert lib().h2r_sha256_hashed_msg_batch(chip._ctx, buf.data_ptr(), None, n, B, None, hashed.data_ptr(), None, 0, st) == 0
    torch.cuda.synchronize()
    d, h = digest.cpu().numpy(), hashed.cpu().numpy().view(np.uint64)
    for i in range(B):
        want = hashlib.sha256(raw[i * n:(i + 1) * n].tobytes()).digest()
        assert d[i].tobytes() == want and np.array_equal(h[i], OL.hashed_msg(want)[0])
    assert lib().h2r_sha256_hashed_msg_batch(chip._ctx, buf.data_ptr(), None, n, B, digest.data_ptr() + 8, None, None, 0, st) == H.H2R_E_SHAPE
    hm = torch.zeros((B, 288), dtype=torch.uint8, device="cuda")
    assert lib().h2r_sha256_hashed_msg_batch(chip._ctx, buf.data_ptr(), None, n, B, None, None, hm.data_ptr(), 272, st) == H.H2R_E_SHAPE
    assert lib().h2r_sha256_hashed_msg_batch(chip._ctx, None, None, n, B, digest.data_ptr(), None, None, 0, st) != 0
    assert lib().h2r_sha256_hashed_msg_batch(chip._ctx, buf.data_ptr(), None, n, 0, digest.data_ptr(), None, None, 0, st) == 0
    # zero-length fixed form: every digest is SHA-256("")
    assert lib().h2r_sha256_hashed_msg_batch(chip._ctx, None, None, 0, 4, digest.data_ptr(), None, None, 0, st) == 0
    torch.cuda.synchronize()
    assert digest[:4].cpu().numpy().tobytes() == hashlib.sha256(b"").digest() * 4


@pytest.mark.gpu
def test_signature_verifier_batch_from_message_bytes(H, golden):
    """h2r_signature_verifier_batch = RSASignatureVerifier::verify_pkcs1v15_signature (src/lib.rs:183-246) on the reference's
    three RSA known-answer signatures of b"hello world" plus freshly signed ragged messages (a 2048-bit key made here: the
    reference's tests sign random 128-byte messages with a random key, src/lib.rs:339-356): is_valid, the returned digest bytes,
    the hashed-message stream and the whole verify stream equal the oracles'."""
    import torch
    kats = golden["rsa_kats"]
    rng = random.Random(11)
    # an RSA key from two fixed probable primes (Miller-Rabin below), e = 65537
    def is_prime(v):
        if v % 2 == 0:
            return v == 2
        d, s = v - 1, 0
        while d % 2 == 0:
            d, s = d // 2, s + 1
        for a in (2, 3, 5, 7, 11, 13, 17, 19, 23, 29, 31, 37):
            x = pow(a, d, v)
            if x in (1, v - 1):
                continue
            for _ in range(s - 1):
                x = x * x % v
                if x == v - 1:
                    break
            else:
                return False
        return True

    def prime(bits):
        while True:
            v = rng.getrandbits(bits) | (1 << (bits - 1)) | (1 << (bits - 2)) | 1
            if v % 65537 != 1 and is_prime(v):
                return v
    p_, q_ = prime(1024), prime(1024)
    n_own = p_ * q_
    assert n_own.bit_length() == 2048
    d_own = pow(65537, -1, (p_ - 1) * (q_ - 1))

    def sign(msg):   # pkcs1v15 with SHA-256 (RFC 8017 9.2: 00 01 ff.. 00 DigestInfo H)
        t = bytes.fromhex("3031300d060960864801650304020105000420") + hashlib.sha256(msg).digest()
        em = b"\x00\x01" + b"\xff" * (256 - 3 - len(t)) + b"\x00" + t
        return pow(int.from_bytes(em, "big"), d_own, n_own)

    own_msgs = [b"", b"a", bytes(rng.getrandbits(8) for _ in range(128)), bytes(rng.getrandbits(8) for _ in range(55)),
                bytes(rng.getrandbits(8) for _ in range(192))]
    msgs = [b"hello world"] * 3 + own_msgs + [b"tampered message", b"hello world"]
    ns = [int(k["n"]) for k in kats] + [n_own] * (len(own_msgs) + 1) + [int(kats[0]["n"])]
    sigs = [int(k["sig"]) for k in kats] + [sign(m) for m in own_msgs] + [sign(b"original message"), int(kats[1]["sig"])]
    want_valid = [1, 1, 0] + [1] * len(own_msgs) + [0, 0]
    rsa = H.RSAChip(2048, 5)
    verifier = H.RSASignatureVerifier(rsa, sha256_max_byte_size=128 + 64)
    pk = rsa.assign_public_key(H.RSAPublicKey(H.UnassignedInteger.from_ints(ns, 32, 64), H.Fix(65537)))
    sg = rsa.assign_signature(H.RSASignature(H.UnassignedInteger.from_ints(sigs, 32, 64)))
    res = verifier.verify_pkcs1v15_signature(pk, msgs, sg)
    torch.cuda.synchronize()
    assert res.status.cpu().tolist() == [0] * len(msgs)
    assert res.is_valid.cpu().tolist() == want_valid
    o = Oracle(64, 32)
    digest, hm = res.digest.cpu().numpy(), res.hashed_msg_trace.cpu().numpy()
    hashed_dev = res.inputs[2].cpu().numpy().view(np.uint64)
    for i, m in enumerate(msgs):
        d = OL.sha256(m)
        limbs, st = OL.hashed_msg(d)
        assert digest[i].tobytes() == d and np.array_equal(hm[i], st) and np.array_equal(hashed_dev[i], limbs)
        _, _, s_if = o.assert_in_field(o.limbs(sigs[i]), o.limbs(ns[i]))
        _, out, s_pow = o.pow_mod_fixed_exp(o.limbs(sigs[i]), o.limbs(ns[i]), 65537)
        _, valid, s_em = o.pkcs1v15_em_check(out, limbs)
        assert valid == want_valid[i]
        assert np.array_equal(res.flatten(i), np.concatenate([s_if, s_pow, s_em])), i
    with pytest.raises(ValueError):
        verifier.verify_pkcs1v15_signature(pk, [bytes(193)] * len(msgs), sg)
    # one message signed by every element (the reference's bench: one msg, src/lib.rs:339)
    res1 = verifier.verify_pkcs1v15_signature(pk, b"hello world", sg)
    torch.cuda.synchronize()
    assert res1.is_valid.cpu().tolist() == [1, 1, 0] + [0] * (len(msgs) - 3)


@pytest.mark.gpu
def test_hashed_msg_advice_rows(H, golden):
    """The limb composition of src/lib.rs:225-239 as advice rows (h2r_hashed_msg_emit_advice): 68 rows per element equal to the
    Python restatement built from the ORACLE's stream, kinds match, every row satisfies the main gate with the C ABI's fixed row;
    the verifier's whole-region image = those rows followed by the h2r_verify_emit_advice rows."""
    import torch
    import advice_ref as AR
    from halo2_rsa_amd import _lib
    from halo2_rsa_amd._lib import lib
    kats = golden["rsa_kats"]
    P = R.FIELD_MODULI["bn254_fr"]
    rsa = H.RSAChip(2048, 5)
    chip = rsa.bigint_chip()
    msgs = [b"hello world", b"", bytes(range(200)), b"hello world"]
    ns = [int(kats[0]["n"])] * 3 + [int(kats[2]["n"])]
    sigs = [int(kats[0]["sig"])] * 3 + [int(kats[2]["sig"])]
    pk = rsa.assign_public_key(H.RSAPublicKey(H.UnassignedInteger.from_ints(ns, 32, 64), H.Fix(65537)))
    sg = rsa.assign_signature(H.RSASignature(H.UnassignedInteger.from_ints(sigs, 32, 64)))
    res = H.RSASignatureVerifier(rsa).verify_pkcs1v15_signature(pk, msgs, sg)
    rows = int(lib().h2r_hashed_msg_advice_rows(chip._ctx))
    assert rows == 4 * (1 + 8 * 2) == 68
    kinds = np.zeros(rows, dtype=np.uint8)
    assert lib().h2r_hashed_msg_row_kinds(chip._ctx, kinds.ctypes.data) == 0
    total, _ = res.advice_sections()
    whole = res.emit_advice(with_hashed_msg=True)
    plain = res.emit_advice()
    torch.cuda.synchronize()
    assert res.is_valid.cpu().tolist() == [1, 0, 0, 0]
    whole = whole.cpu().numpy().reshape(len(msgs), rows + total, 160)
    assert np.array_equal(whole[:, rows:], plain.cpu().numpy().reshape(len(msgs), total, 160))
    for i, m in enumerate(msgs):
        limbs, st = OL.hashed_msg(OL.sha256(m))
        im, ref_limbs = AR.hashed_msg_image(st, P)
        assert ref_limbs == [int(v) for v in limbs] and im.kinds == kinds.tolist()
        got = whole[i, :rows]
        assert np.array_equal(got, AR.image_bytes(im)), i
        cells = [[int.from_bytes(got[r, 32 * c:32 * c + 32].tobytes(), "little") for c in range(5)] for r in range(rows)]
        for r in range(rows):
            fr = _lib.H2RFixedRow()
            assert lib().h2r_advice_fixed_row(chip._ctx, None, int(kinds[r]), ctypes.byref(fr)) == 0
            f = fr.as_dict()
            ref = AR.fixed_row(int(kinds[r]), 64, 32, 0, 0, 0, None)
            assert {nm: v % P for nm, v in ref.items() if nm in AR.FIXED_NAMES} == {nm: f[nm] for nm in AR.FIXED_NAMES}, (r, kinds[r])
            assert AR.gate_residual(cells[r], 0, f, P) == 0, (r, kinds[r])
        # the last limb_val of each limb is the cell handed to RSAChip::verify_pkcs1v15_signature (src/lib.rs:237-241)
        assert [cells[17 * k + 16][3] for k in range(4)] == ref_limbs


@pytest.mark.gpu
def test_verify_with_a_variable_exponent(H, golden):
    """RSAChip::verify_pkcs1v15_signature / RSASignatureVerifier with RSAPubE::Var (src/chip.rs:108-110: pow_mod with the chip's
    exp_limb_bits): h2r_verify_pkcs1v15_var_batch on the reference's KATs with e = 65537 given as ONE 17-bit exponent limb and as
    four 5-bit limbs (the reference's EXP_LIMB_BITS = 5, src/chip.rs:364) -- is_valid 1, 1, 0; the element's stream = the oracle's
    in-field + variable-exponent pow + encoded-message streams; an exponent limb wider than exp_limb_bits gets H2R_E_SHAPE; the
    whole-element advice image holds the pow_mod rows (tests/test_cells_direct.py checks them cell by cell)."""
    import torch
    from halo2_rsa_amd._lib import lib
    kats = golden["rsa_kats"]
    ns = [int(k["n"]) for k in kats]
    sigs = [int(k["sig"]) for k in kats]
    hashed = [int(k["hashed"]) for k in kats]
    o = Oracle(64, 32)
    for exp_limb_bits, e_limbs in ((17, [65537]), (5, [1, 0, 0, 2])):       # 65537 = 1 + 2 * 32^3
        assert sum(v << (exp_limb_bits * i) for i, v in enumerate(e_limbs)) == 65537
        rsa = H.RSAChip(2048, exp_limb_bits)
        e_un = H.UnassignedInteger(np.array([e_limbs] * 3, dtype=np.uint64))
        pk = rsa.assign_public_key(H.RSAPublicKey(H.UnassignedInteger.from_ints(ns, 32, 64), H.Var(e_un)))
        sg = rsa.assign_signature(H.RSASignature(H.UnassignedInteger.from_ints(sigs, 32, 64)))
        res = rsa.verify_pkcs1v15_signature(pk, hashed, sg)
        res_m = H.RSASignatureVerifier(rsa).verify_pkcs1v15_signature(pk, b"hello world", sg)
        torch.cuda.synchronize()
        for r in (res, res_m):
            assert r.status.cpu().tolist() == [0, 0, 0] and r.is_valid.cpu().tolist() == [1, 1, 0]
        nbits = exp_limb_bits * len(e_limbs)                              # [is_eq] [assert_in_field] [to_bits, acc = 1, per bit mul_mod / select / square_mod] [EM]
        assert res.advice_sections()[1][2] == len(e_limbs) * (exp_limb_bits + (exp_limb_bits + 3) // 4 + 1) + 2 + nbits * (2 * 3974 + 32)
        for i in range(3):
            _, _, s_if = o.assert_in_field(o.limbs(sigs[i]), o.limbs(ns[i]))
            rc, out, s_pow = o.pow_mod(o.limbs(sigs[i]), np.array(e_limbs, dtype=np.uint64), exp_limb_bits, o.limbs(ns[i]))
            assert rc == 0 and R.from_limbs([int(v) for v in out], 64) == pow(sigs[i], 65537, ns[i])
            _, valid, s_em = o.pkcs1v15_em_check(out, o.limbs(hashed[i], 4))
            want = np.concatenate([s_if, s_pow, s_em])
            assert np.array_equal(res.flatten(i), want) and np.array_equal(res_m.flatten(i), want), (exp_limb_bits, i)
    # the pipelined form (h2r_pipeline_verify_pkcs1v15_var): three calls over two buffer sets give the batch export's element bytes
    import ctypes
    rsa = H.RSAChip(2048, 5)
    chip = rsa.bigint_chip()
    pk = rsa.assign_public_key(H.RSAPublicKey(H.UnassignedInteger.from_ints(ns, 32, 64), H.Var(H.UnassignedInteger(np.array([[1, 0, 0, 2]] * 3, dtype=np.uint64)))))
    sg = rsa.assign_signature(H.RSASignature(H.UnassignedInteger.from_ints(sigs, 32, 64)))
    ref = rsa.verify_pkcs1v15_signature(pk, hashed, sg)
    vl = ref.layout
    hashed_dev = ref.inputs[2]
    pipe = H.Pipeline(chip, depth=2)
    bufs = [dict(trace=torch.zeros(3 * vl.elem_stride, dtype=torch.uint8, device="cuda"), powed=torch.zeros((3, 32), dtype=torch.int64, device="cuda"),
                 valid=torch.zeros(3, dtype=torch.uint8, device="cuda"), status=torch.zeros(3, dtype=torch.uint8, device="cuda"),
                 ws=torch.zeros(chip.workspace_bytes(3, vl.pow.num_mul_mods), dtype=torch.uint8, device="cuda")) for _ in range(2)]
    for k in range(3):
        b = bufs[k & 1]
        pipe.verify_pkcs1v15_var(sg.c, pk.e.e, 5, pk.n, hashed_dev, b["trace"], b["ws"], b["powed"], b["valid"], b["status"])
    pipe.join()
    torch.cuda.synchronize()
    for b in bufs:
        assert b["valid"].cpu().tolist() == [1, 1, 0] and b["status"].cpu().tolist() == [0, 0, 0]
        assert torch.equal(b["powed"], ref.powed.limbs_dev)
        got = H.rsa.VerifyResult(b["valid"], H.AssignedInteger(b["powed"], 64), b["status"], b["trace"], vl, chip)
        for i in range(3):
            assert np.array_equal(got.flatten(i), ref.flatten(i)), i
    pipe.close()
    # the same at 1,024 signatures per call (one-launch steps: the chain role of the verifier's build writes the witness): every verdict,
    # status and result equals the batch export's, sampled element bytes too
    B = 1024
    nsB = [ns[i % 3] for i in range(B)]
    sgB = [sigs[i % 3] if i % 7 else (sigs[i % 3] ^ 4) for i in range(B)]
    hsB = [hashed[i % 3] for i in range(B)]
    pkB = rsa.assign_public_key(H.RSAPublicKey(H.UnassignedInteger.from_ints(nsB, 32, 64), H.Var(H.UnassignedInteger(np.array([[1, 0, 0, 2]] * B, dtype=np.uint64)))))
    sB = rsa.assign_signature(H.RSASignature(H.UnassignedInteger.from_ints(sgB, 32, 64)))
    refB = rsa.verify_pkcs1v15_signature(pkB, hsB, sB)
    vlB = refB.layout
    pipe = H.Pipeline(chip, depth=2)
    bufsB = [dict(trace=torch.zeros(B * vlB.elem_stride, dtype=torch.uint8, device="cuda"), powed=torch.zeros((B, 32), dtype=torch.int64, device="cuda"),
                  valid=torch.zeros(B, dtype=torch.uint8, device="cuda"), status=torch.zeros(B, dtype=torch.uint8, device="cuda"),
                  ws=torch.zeros(chip.workspace_bytes(B, vlB.pow.num_mul_mods), dtype=torch.uint8, device="cuda")) for _ in range(2)]
    for k in range(3):
        b = bufsB[k & 1]
        pipe.verify_pkcs1v15_var(sB.c, pkB.e.e, 5, pkB.n, refB.inputs[2], b["trace"], b["ws"], b["powed"], b["valid"], b["status"])
    pipe.join()
    torch.cuda.synchronize()
    for b in bufsB:
        assert torch.equal(b["valid"], refB.is_valid) and torch.equal(b["status"], refB.status) and torch.equal(b["powed"], refB.powed.limbs_dev)
        got = H.rsa.VerifyResult(b["valid"], H.AssignedInteger(b["powed"], 64), b["status"], b["trace"], vlB, chip)
        for i in (0, 1, 2, 7, 1023):
            assert np.array_equal(got.flatten(i), refB.flatten(i)), i
    assert refB.is_valid[:3].cpu().tolist() == [0, 1, 0] and int(refB.is_valid.sum()) > 500
    pipe.close()
    # a limb that does not fit exp_limb_bits: main_gate.to_bits cannot be satisfied (big_integer/chip.rs:677)
    rsa = H.RSAChip(2048, 5)
    pk = rsa.assign_public_key(H.RSAPublicKey(H.UnassignedInteger.from_ints(ns, 32, 64), H.Var(H.UnassignedInteger(np.array([[1, 0, 0, 32]] * 3, dtype=np.uint64)))))
    res = rsa.verify_pkcs1v15_signature(pk, hashed, rsa.assign_signature(H.RSASignature(H.UnassignedInteger.from_ints(sigs, 32, 64))))
    torch.cuda.synchronize()
    assert res.status.cpu().tolist() == [H.H2R_E_SHAPE] * 3 and res.is_valid.cpu().tolist() == [0, 0, 0]


@pytest.mark.gpu
def test_pipelined_signature_verifier_sha_role(H, golden):
    """h2r_pipeline_signature_verifier: calls of 1,024 signatures (one-launch steps: the SHA-256 / hashed-message role rides inside the
    step launch from the second call on; the first call of the train hashes in a kernel of its own) and calls of 3 signatures
    (two-queue form) over two buffer sets, ragged messages with a different length mix per call: every digest / limb set / stream
    equals the C oracle's, is_valid and the element bytes of sampled signatures equal the batch export's; the messages of a call
    may be overwritten as soon as the call returns (they are read in stream order inside it)."""
    import torch
    kats = golden["rsa_kats"]
    rng = random.Random(21)
    rsa = H.RSAChip(2048, 5)
    chip = rsa.bigint_chip()
    for B in (1024, 3, 5000):   # 5,000: walked as two step launches (2,560 + 2,440), the SHA role of the whole call in the first
        ns = [int(kats[i % 3]["n"]) for i in range(B)]
        sigs = [int(kats[i % 3]["sig"]) for i in range(B)]
        pk = rsa.assign_public_key(H.RSAPublicKey(H.UnassignedInteger.from_ints(ns, 32, 64), H.Fix(65537)))
        sg = rsa.assign_signature(H.RSASignature(H.UnassignedInteger.from_ints(sigs, 32, 64)))
        vl = rsa._verify_layout(pk)
        pipe = H.Pipeline(chip, depth=2)
        sets = [dict(trace=torch.zeros(B * vl.elem_stride, dtype=torch.uint8, device="cuda"), powed=torch.zeros((B, 32), dtype=torch.int64, device="cuda"),
                     valid=torch.zeros(B, dtype=torch.uint8, device="cuda"), status=torch.zeros(B, dtype=torch.uint8, device="cuda"),
                     hashed=torch.zeros((B, 4), dtype=torch.int64, device="cuda"), digest=torch.zeros((B, 32), dtype=torch.uint8, device="cuda"),
                     hm=torch.zeros((B, 288), dtype=torch.uint8, device="cuda"),
                     ws=torch.zeros(chip.workspace_bytes(B, vl.pow.num_mul_mods), dtype=torch.uint8, device="cuda")) for _ in range(2)]
        staging = torch.zeros(B * 200 + 16, dtype=torch.uint8, device="cuda")     # ONE message staging buffer, refilled per call
        off_dev = torch.zeros(B + 1, dtype=torch.int64, device="cuda")
        calls = []
        for k in range(4):
            msgs = [b"hello world" if (i % 3 != 2 and (i + k) % 5) else bytes(rng.getrandbits(8) for _ in range(rng.choice([0, 1, 55, 56, 64, 119, 128, 190])))
                    for i in range(B)]
            buf, off = H.pack_messages(msgs, torch.device("cuda", 0))
            staging[:buf.numel()].copy_(buf)            # stream-ordered refill of the staging buffers, right behind the previous call
            off_dev.copy_(off)
            b = sets[k & 1]
            pipe.signature_verifier(staging, off_dev, 0, sg.c, 65537, pk.n, b["trace"], b["ws"], b["powed"], b["valid"], b["status"], b["hashed"],
                                    b["digest"], b["hm"])
            calls.append(msgs)
            if k >= 2:                                   # check the set this call used before it is reused (k + 2)
                pass
        pipe.join()
        torch.cuda.synchronize()
        for k in (2, 3):                                 # the last user of each buffer set
            b, msgs = sets[k & 1], calls[k]
            digest, hm, hashed = b["digest"].cpu().numpy(), b["hm"].cpu().numpy(), b["hashed"].cpu().numpy().view(np.uint64)
            valid = b["valid"].cpu().tolist()
            assert b["status"].cpu().tolist() == [0] * B
            for i in range(B):
                d = OL.sha256(msgs[i])
                limbs, st = OL.hashed_msg(d)
                assert digest[i].tobytes() == d and np.array_equal(hashed[i], limbs) and np.array_equal(hm[i], st), (B, k, i)
                assert valid[i] == (1 if (msgs[i] == b"hello world" and i % 3 != 2) else 0), (B, k, i)
            ref = rsa.verify_pkcs1v15_signature(pk, b["hashed"], sg)
            torch.cuda.synchronize()
            got = H.rsa.VerifyResult(b["valid"], H.AssignedInteger(b["powed"], 64), b["status"], b["trace"], vl, chip)
            for i in ([0, 1, 2, 511, B - 1] + ([2559, 2560, 2561] if B == 5000 else []) if B >= 1024 else [0, 1, 2]):
                assert np.array_equal(got.flatten(i), ref.flatten(i)), (B, k, i)
        pipe.close()


@pytest.mark.gpu
def test_packed_messages_are_validated(H, golden):
    """A caller-packed (buffer, offsets) pair is checked before the kernels take message e from off[e] to off[e + 1]: wrong
    length / dtype / device, decreasing offsets, an end behind the buffer, and a message above sha256_max_byte_size are refused;
    a well-formed pair gives the same verdicts as the list form."""
    import torch
    import halo2_rsa_amd.rsa as RS
    rsa = H.RSAChip(2048, 5)
    kats = golden["rsa_kats"]
    ns, sigs = [int(k["n"]) for k in kats], [int(k["sig"]) for k in kats]
    pk = rsa.assign_public_key(H.RSAPublicKey(H.UnassignedInteger.from_ints(ns, 32, 64), H.Fix(65537)))
    sg = rsa.assign_signature(H.RSASignature(H.UnassignedInteger.from_ints(sigs, 32, 64)))
    ver = H.RSASignatureVerifier(rsa, sha256_max_byte_size=64)
    msgs = [b"hello world"] * 3
    want = ver.verify_pkcs1v15_signature(pk, msgs, sg).is_valid.cpu().tolist()
    buf, off = RS.pack_messages(msgs, torch.device("cuda", 0))
    assert ver.verify_pkcs1v15_signature(pk, (buf, off), sg).is_valid.cpu().tolist() == want == [1, 1, 0]
    bad = [(buf, off[:-1].contiguous()), (buf, off.to(torch.int32)), (buf, off.cpu()), (buf, torch.flip(off, [0]).contiguous()),
           (buf, off + buf.numel()), (buf[:5].contiguous(), off)]
    for pair in bad:
        with pytest.raises((ValueError, TypeError)):
            ver.verify_pkcs1v15_signature(pk, pair, sg)
    big = RS.pack_messages([b"x" * 65, b"y", b"z"], torch.device("cuda", 0))
    with pytest.raises(ValueError):
        ver.verify_pkcs1v15_signature(pk, big, sg)
