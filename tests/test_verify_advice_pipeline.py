"""h2r_pipeline_verify_pkcs1v15_advice: the whole RSAChip::verify_pkcs1v15_signature element (src/chip.rs:128-199 -- is_eq seed :137,
assert_in_field :106, pow_mod_fixed_exp :111, the encoded-message check :138-198) as advice rows WITHOUT records, pipelined.  The
reference's own circuits for this method are TestRSASignatureCircuit1 / 2 and the BAD variant (src/chip.rs:694-838): the KAT1 / KAT2 /
BAD vectors are elements here.  The record-based image (h2r_verify_pkcs1v15_batch + h2r_verify_emit_advice) is pinned against the Python
restatement run on the oracle's stream (tests/test_gpu_parity.py); the records-free pipelined image must be that image, byte for byte,
in every representation, and must pass the device-side MockProver (h2r_advice_check)."""
import ctypes
import os
import random
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))


def test_compact_layout_on_the_host():
    """h2r_verify_layout_compact: the pow layout untouched, the witness regions packed from offset 0, a 256-byte stride; refused where
    RSAChip is (limb width 64 only, src/chip.rs:203)."""
    from halo2_rsa_amd import _lib
    from halo2_rsa_amd._lib import lib
    ctx = ctypes.c_void_p()
    p = _lib.H2RParams(64, 2048, 0, -1)
    assert lib().h2r_ctx_create(ctypes.byref(p), ctypes.byref(ctx)) == 0
    e = (65537).to_bytes(3, "little")
    full, vl = _lib.H2RVerifyLayout(), _lib.H2RVerifyLayout()
    assert lib().h2r_verify_layout_fixed(ctx, e, len(e), ctypes.byref(full)) == 0
    assert lib().h2r_verify_layout_compact(ctx, ctypes.byref(full), ctypes.byref(vl)) == 0
    assert vl.pow.off_records == 2 ** 64 - 1          # no record planes: what the record exports look at
    vl.pow.off_records = full.pow.off_records
    assert bytes(vl.pow) == bytes(full.pow)            # ... everything else of the pow layout untouched
    vl.pow.off_records = 2 ** 64 - 1
    assert vl.off_in_field == 0 and vl.off_em == full.off_em - full.off_in_field
    assert vl.elem_stride % 256 == 0 and vl.elem_stride >= vl.off_em + full.em_stream_bytes and vl.off_em >= full.in_field_stream_bytes
    assert vl.elem_stride < 16384 < full.elem_stride
    assert (vl.in_field_stream_bytes, vl.em_stream_bytes, vl.stream_bytes) == (full.in_field_stream_bytes, full.em_stream_bytes, full.stream_bytes)
    sa, sb = (ctypes.c_uint64 * 4)(), (ctypes.c_uint64 * 4)()
    assert lib().h2r_verify_advice_rows(ctx, ctypes.byref(vl), sa) == lib().h2r_verify_advice_rows(ctx, ctypes.byref(full), sb) == 77219
    assert list(sa) == list(sb) == [1, 1532, 75508, 178]
    assert lib().h2r_verify_layout_compact(ctx, None, ctypes.byref(vl)) == _lib.H2R_E_NULL
    lib().h2r_ctx_destroy(ctx)
    p32 = _lib.H2RParams(32, 2048, 0, -1)
    assert lib().h2r_ctx_create(ctypes.byref(p32), ctypes.byref(ctx)) == 0
    assert lib().h2r_verify_layout_compact(ctx, ctypes.byref(full), ctypes.byref(vl)) == _lib.H2R_E_SHAPE
    lib().h2r_ctx_destroy(ctx)


torch = pytest.importorskip("torch")


@pytest.fixture(scope="module")
def H():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    import halo2_rsa_amd as H_
    return H_


def rand_modulus(rng, bits):
    return rng.getrandbits(bits) | (1 << (bits - 1)) | 1


def _hashed_tensor(vals):
    limbs = [[(h >> (64 * j)) & 0xFFFFFFFFFFFFFFFF for j in range(4)] for h in vals]
    return torch.tensor(np.array(limbs, dtype=np.uint64).view(np.int64), device="cuda")


def _buffers(chip, vl, rows, B):
    return dict(ws=torch.empty(chip.workspace_bytes(B, vl.pow.num_mul_mods), dtype=torch.uint8, device="cuda"),
                wit=torch.zeros((B, vl.elem_stride), dtype=torch.uint8, device="cuda"),
                powed=torch.zeros((B, chip.num_limbs), dtype=torch.int64, device="cuda"),
                valid=torch.zeros(B, dtype=torch.uint8, device="cuda"), st=torch.zeros(B, dtype=torch.uint8, device="cuda"),
                img=torch.empty((B, chip.image_bytes(rows)), dtype=torch.uint8, device="cuda"))


def _inputs(golden, rng, B, bad_elem=None):
    """KAT1, KAT2, BAD (src/chip.rs:703-798) + random moduli / signatures (their encoded message is wrong: is_valid = 0, every gate holds)."""
    kats = golden["rsa_kats"]
    ns = [int(k["n"]) for k in kats] + [rand_modulus(rng, 2048) for _ in range(B - 3)]
    sigs = [int(k["sig"]) for k in kats] + [rng.randrange(n) for n in ns[3:]]
    hashed = [int(k["hashed"]) for k in kats] + [rng.getrandbits(256) for _ in range(B - 3)]
    if bad_elem is not None:
        sigs[bad_elem] = ns[bad_elem] + 7          # not in the field: assert_in_field fails, the element has a status and no image
    return ns, sigs, hashed


@pytest.mark.gpu
@pytest.mark.parametrize("repr_kw", [dict(), dict(columns=True), dict(montgomery=True), dict(columns=True, montgomery=True)])
def test_pipelined_verify_image_is_the_record_based_image(H, golden, repr_kw):
    """Four pipelined calls over two buffer sets (different inputs per call, one element not in the field): powed, is_valid, status and
    every byte of every image equal what h2r_verify_pkcs1v15_batch + h2r_verify_emit_advice give for the same inputs; an element with a
    status keeps its bytes; the plain emit reads the compact witness as its `trace` (H2R_ADVICE_DIRECT) and gives the same image."""
    from halo2_rsa_amd._lib import lib
    rsa = H.RSAChip(2048, 5, **repr_kw)
    chip = rsa.bigint_chip()
    rng = random.Random(0x68327273 + 61)
    B, depth, calls_n = 7, 2, 4
    pipe = H.Pipeline(chip, depth, 2)
    vl = pipe.verify_compact_layout(65537)
    sec = (ctypes.c_uint64 * 4)()
    rows = int(lib().h2r_verify_advice_rows(chip._ctx, ctypes.byref(vl), sec))
    assert rows == 77219
    sets = [_buffers(chip, vl, rows, B) for _ in range(depth)]
    calls, got = [], []
    for k in range(calls_n):
        ns, sigs, hashed = _inputs(golden, rng, B, bad_elem=5 if k == 1 else None)
        calls.append((ns, sigs, hashed))
        s = sets[k % depth]
        if k >= depth:
            got.append((k - depth, {key: v.clone() for key, v in s.items() if key != "ws"}))
        s["img"].fill_(0x5A)
        pipe.verify_pkcs1v15_advice(chip.assign_integer(sigs), 65537, chip.assign_integer(ns), _hashed_tensor(hashed), s["wit"], s["ws"],
                                    s["powed"], s["valid"], s["st"], s["img"])
    pipe.join()
    for k in range(calls_n - depth, calls_n):
        got.append((k, {key: v.clone() for key, v in sets[k % depth].items() if key != "ws"}))
    pipe.close()
    torch.cuda.synchronize()
    assert sorted(g[0] for g in got) == list(range(calls_n))
    for k, g in got:
        ns, sigs, hashed = calls[k]
        pk = rsa.assign_public_key(H.RSAPublicKey(H.UnassignedInteger.from_ints(ns, 32, 64), H.Fix(65537)))
        sg = rsa.assign_signature(H.RSASignature(H.UnassignedInteger.from_ints(sigs, 32, 64)))
        res = rsa.verify_pkcs1v15_signature(pk, _hashed_tensor(hashed), sg)
        want = res.emit_advice()
        assert torch.equal(g["st"], res.status) and torch.equal(g["valid"], res.is_valid), k
        assert g["valid"].cpu().tolist()[:3] == [1, 1, 0]
        ok = (res.status == 0)
        assert torch.equal(g["powed"][ok], res.powed.limbs_dev[ok]), k
        assert torch.equal(g["img"][ok], want[ok]), k
        if k == 1:
            assert int(g["st"][5]) == H.H2R_E_NOT_IN_FIELD and int(g["valid"][5]) == 0 and bool((g["img"][5] == 0x5A).all())
        # the same layout through the plain emit: the witness as `trace`, H2R_ADVICE_DIRECT
        if k == calls_n - 1:
            out = torch.full_like(g["img"], 0x5A)
            s = sets[k % depth]
            from halo2_rsa_amd import _lib
            sig_d, n_d, h_d = chip.assign_integer(sigs), chip.assign_integer(ns), _hashed_tensor(hashed)
            assert lib().h2r_verify_emit_advice(chip._ctx, ctypes.byref(vl), sig_d.data_ptr(), n_d.data_ptr(), h_d.data_ptr(), s["powed"].data_ptr(),
                                                _lib.H2R_ADVICE_DIRECT, s["wit"].data_ptr(), s["ws"].data_ptr(), B, s["st"].data_ptr(), out.data_ptr(),
                                                out.shape[1], chip._stream()) == 0
            torch.cuda.synchronize()
            assert torch.equal(out, g["img"])


@pytest.mark.gpu
@pytest.mark.parametrize("repr_kw", [dict(), dict(columns=True, montgomery=True)])
def test_pipelined_verify_image_passes_the_device_mockprover(H, golden, repr_kw):
    """h2r_advice_check on the records-free image: gate + lookup (RangeChip's 4-bit table included) + the pow rows' copy pairs, every
    element; a flipped cell in the encoded-message section and one in the pow section are found."""
    from halo2_rsa_amd._lib import lib
    rsa = H.RSAChip(2048, 5, **repr_kw)
    chip = rsa.bigint_chip()
    look = H.LookupArgument(chip, rsa_chip=True)
    rng = random.Random(17)
    B = 6
    pipe = H.Pipeline(chip, 2, 2)
    vl = pipe.verify_compact_layout(65537)
    sec = (ctypes.c_uint64 * 4)()
    rows = int(lib().h2r_verify_advice_rows(chip._ctx, ctypes.byref(vl), sec))
    s = _buffers(chip, vl, rows, B)
    ns, sigs, hashed = _inputs(golden, rng, B)
    sig_d, n_d = chip.assign_integer(sigs), chip.assign_integer(ns)
    pipe.verify_pkcs1v15_advice(sig_d, 65537, n_d, _hashed_tensor(hashed), s["wit"], s["ws"], s["powed"], s["valid"], s["st"], s["img"])
    pipe.join()
    pipe.close()
    kinds = np.zeros(rows, dtype=np.uint8)
    assert lib().h2r_verify_row_kinds(chip._ctx, ctypes.byref(vl), kinds.ctypes.data) == 0
    copies = chip.pow_copy_map(vl.pow, 65537, row_offset=sec[0] + sec[1])
    bad, first = chip.advice_check(kinds, s["img"], B, copies=copies, src_a=sig_d, src_n=n_d, lookup=look)
    assert bad.cpu().tolist() == [0] * B, (bad.cpu().tolist(), [hex(v) for v in first.cpu().tolist()])
    assert s["valid"].cpu().tolist() == [1, 1, 0, 0, 0, 0]
    if not repr_kw:
        img = s["img"].clone()
        em0 = sec[0] + sec[1] + sec[2]
        r_em = next(r for r in range(em0, rows) if kinds[r] == 48)      # H2R_ROW_RANGE_U32: a sub-limb that leaves the table
        img[2, r_em * 160 + 20] ^= 1
        r_pow = sec[0] + sec[1] + 2 + 128 + 4                            # record 0, column 1's second multiply-add: its acc cell
        img[4, r_pow * 160 + 96] ^= 1
        bad, first = chip.advice_check(kinds, img, B, copies=copies, src_a=sig_d, src_n=n_d, lookup=look)
        b = bad.cpu().tolist()
        assert b[2] > 0 and b[4] > 0 and b[0] == b[1] == b[3] == b[5] == 0, b


# ---- RSAPubE::Var without records (h2r_pipeline_modpow_public_key_var_advice) ----------------------------------------------------------
def test_pow_compact_layout_on_the_host():
    """h2r_pow_layout_compact: the counts untouched, no record planes (off_records = UINT64_MAX), the Var element's selected operands /
    result / exponent bits packed from offset 0; a Fix layout keeps its result only."""
    from halo2_rsa_amd import _lib
    from halo2_rsa_amd._lib import lib
    ctx = ctypes.c_void_p()
    p = _lib.H2RParams(64, 2048, 0, -1)
    assert lib().h2r_ctx_create(ctypes.byref(p), ctypes.byref(ctx)) == 0
    full, pl = _lib.H2RPowLayout(), _lib.H2RPowLayout()
    assert lib().h2r_pow_var_layout(ctx, 1, 5, ctypes.byref(full)) == 0
    assert lib().h2r_pow_layout_compact(ctx, ctypes.byref(full), ctypes.byref(pl)) == 0
    assert (pl.num_mul_mods, pl.num_exp_bits, pl.exp_limb_bits, pl.e_num_limbs, pl.stream_bytes) == (10, 5, 5, 1, full.stream_bytes)
    assert pl.off_records == 2 ** 64 - 1 and pl.off_selected == 0 and pl.selected_stride == 256
    assert pl.off_result == 5 * 256 and pl.off_e_bits == 6 * 256 and pl.elem_stride == 7 * 256 < full.elem_stride
    assert lib().h2r_pow_advice_rows(ctx, ctypes.byref(pl)) == lib().h2r_pow_advice_rows(ctx, ctypes.byref(full))
    e = (65537).to_bytes(3, "little")
    assert lib().h2r_pow_fixed_layout(ctx, e, len(e), ctypes.byref(full)) == 0
    assert lib().h2r_pow_layout_compact(ctx, ctypes.byref(full), ctypes.byref(pl)) == 0
    assert pl.off_e_bits == 2 ** 64 - 1 and pl.off_result == 0 and pl.elem_stride == 256 and pl.num_mul_mods == 19
    assert lib().h2r_pow_layout_compact(ctx, None, ctypes.byref(pl)) == _lib.H2R_E_NULL
    lib().h2r_ctx_destroy(ctx)


@pytest.mark.gpu
@pytest.mark.parametrize("repr_kw", [dict(), dict(columns=True, montgomery=True)])
def test_pipelined_var_arm_image_is_the_record_based_image(H, repr_kw):
    """RSAChip::modpow_public_key with RSAPubE::Var (the arm the reference's TestRSASignatureCircuit2 takes: src/chip.rs:283, 327 use a 5-bit
    e) as advice rows without records: three pipelined calls over two buffer sets, per-element exponents (all-zero among them), one element
    not in the field -- result, status and every byte of every image equal h2r_modpow_public_key_var_batch + h2r_modpow_public_key_emit_advice
    (pinned against the Python restatement on the oracle's Var stream in tests/test_cells_direct.py); h2r_advice_check passes on it."""
    from halo2_rsa_amd import _lib
    from halo2_rsa_amd._lib import lib
    nb, n_el = 5, 2
    rsa = H.RSAChip(2048, nb, **repr_kw)
    chip = rsa.bigint_chip()
    rng = random.Random(77)
    B, depth, calls_n = 5, 2, 3
    pipe = H.Pipeline(chip, depth, 2)
    pl = pipe.pow_var_compact_layout(n_el, nb)
    sec = (ctypes.c_uint64 * 2)()
    rows = int(lib().h2r_modpow_public_key_advice_rows(chip._ctx, ctypes.byref(pl), sec))
    ifs = chip.in_field_layout()[0]
    sets = [dict(ws=torch.empty(chip.workspace_bytes(B, pl.num_mul_mods), dtype=torch.uint8, device="cuda"),
                 wit=torch.zeros((B, pl.elem_stride), dtype=torch.uint8, device="cuda"), inf=torch.zeros(B * ifs, dtype=torch.uint8, device="cuda"),
                 out=torch.zeros((B, chip.num_limbs), dtype=torch.int64, device="cuda"), st=torch.zeros(B, dtype=torch.uint8, device="cuda"),
                 img=torch.empty((B, chip.image_bytes(rows)), dtype=torch.uint8, device="cuda")) for _ in range(depth)]
    calls, got = [], []
    for k in range(calls_n):
        N = [rand_modulus(rng, 2048) for _ in range(1 if k == 2 else B)]          # (the last call: ONE modulus for the whole batch)
        X = [rng.randrange(N[0] if k == 2 else N[i]) for i in range(B)]
        E = [[rng.getrandbits(nb) for _ in range(n_el)] for _ in range(B)]
        E[1] = [0] * n_el
        E[2] = [0b10001, 0b00010]
        if k == 1:
            X[3] = N[3] + 2
        calls.append((X, E, N))
        s = sets[k % depth]
        if k >= depth:
            got.append((k - depth, {key: v.clone() for key, v in s.items() if key != "ws"}))
        s["img"].fill_(0x5A)
        e_dev = chip.assign_integer(H.UnassignedInteger(np.array(E, dtype=np.uint64)))
        pipe.modpow_public_key_var_advice(chip.assign_integer(X), e_dev, nb, chip.assign_integer(N), s["ws"], s["out"], s["st"], s["inf"], s["wit"], s["img"])
    pipe.join()
    for k in range(calls_n - depth, calls_n):
        got.append((k, {key: v.clone() for key, v in sets[k % depth].items() if key != "ws"}))
    pipe.close()
    torch.cuda.synchronize()
    look = H.LookupArgument(chip, rsa_chip=True)
    for k, g in got:
        X, E, N = calls[k]
        pkv = rsa.assign_public_key(H.RSAPublicKey(H.UnassignedInteger.from_ints(N, 32, 64), H.Var(H.UnassignedInteger(np.array(E, dtype=np.uint64)))))
        rv = rsa.modpow_public_key(chip.assign_integer(X), pkv)
        want = rv.emit_modpow_advice()
        assert torch.equal(g["st"], rv.status), k
        ok = (rv.status == 0)
        assert want.shape == g["img"].shape and torch.equal(g["img"][ok], want[ok]), k
        if k == 1:
            assert int(g["st"][3]) == H.H2R_E_NOT_IN_FIELD and bool((g["img"][3] == 0x5A).all())
        for i in range(B):
            if int(g["st"][i]) == 0:
                e_int = sum(v << (nb * j) for j, v in enumerate(E[i]))
                got_int = int.from_bytes(g["out"][i].cpu().numpy().tobytes(), "little")
                assert got_int == (pow(X[i], e_int, N[i % len(N)] if len(N) > 1 else N[0]) if e_int else 1), (k, i)     # big_pow_mod returns 1 for e = 0 (utils.rs:2-17)
        # the device MockProver on the records-free image
        k_if = chip.fresh_op_row_kinds(_lib.FRESH_OPS.index("is_in_field"), assert_one=True)
        k_pow = np.zeros(int(sec[1]), dtype=np.uint8)
        assert lib().h2r_pow_row_kinds(chip._ctx, ctypes.byref(pl), k_pow.ctypes.data) == 0
        bad, first = chip.advice_check(np.concatenate([k_if, k_pow]), g["img"], B, status=g["st"], lookup=look)
        assert bad.cpu().tolist() == [0] * B, (k, bad.cpu().tolist(), [hex(v) for v in first.cpu().tolist()])


@pytest.mark.gpu
@pytest.mark.parametrize("repr_kw", [dict(), dict(columns=True, montgomery=True)])
def test_pipelined_verify_var_arm_image_is_the_record_based_image(H, golden, repr_kw):
    """The whole verify_pkcs1v15_signature element with an RSAPubE::Var key (e = 65537 as 17-bit limbs for the KATs -- valid, valid, BAD --
    and random exponents for random signatures), no records: is_valid, powed, status and every byte of the image equal
    h2r_verify_pkcs1v15_var_batch + h2r_verify_emit_advice; the compact layout keeps the pow_mod witness behind the EM region."""
    from halo2_rsa_amd import _lib
    from halo2_rsa_amd._lib import lib
    nb, n_el = 17, 1
    rsa = H.RSAChip(2048, nb, **repr_kw)
    chip = rsa.bigint_chip()
    rng = random.Random(91)
    B, depth, calls_n = 5, 2, 3
    pipe = H.Pipeline(chip, depth, 2)
    vl = pipe.verify_compact_layout_var(n_el, nb)
    assert vl.pow.off_records == 2 ** 64 - 1 and vl.pow.off_selected >= vl.off_em and vl.pow.elem_stride == vl.elem_stride
    sec = (ctypes.c_uint64 * 4)()
    rows = int(lib().h2r_verify_advice_rows(chip._ctx, ctypes.byref(vl), sec))
    sets = [_buffers(chip, vl, rows, B) for _ in range(depth)]
    calls, got = [], []
    for k in range(calls_n):
        ns, sigs, hashed = _inputs(golden, rng, B, bad_elem=4 if k == 1 else None)
        E = [[65537]] * 3 + [[rng.getrandbits(nb)] for _ in range(B - 3)]
        calls.append((ns, sigs, hashed, E))
        s = sets[k % depth]
        if k >= depth:
            got.append((k - depth, {key: v.clone() for key, v in s.items() if key != "ws"}))
        s["img"].fill_(0x5A)
        e_dev = chip.assign_integer(H.UnassignedInteger(np.array(E, dtype=np.uint64)))
        pipe.verify_pkcs1v15_var_advice(chip.assign_integer(sigs), e_dev, nb, chip.assign_integer(ns), _hashed_tensor(hashed), s["wit"], s["ws"],
                                        s["powed"], s["valid"], s["st"], s["img"])
    pipe.join()
    for k in range(calls_n - depth, calls_n):
        got.append((k, {key: v.clone() for key, v in sets[k % depth].items() if key != "ws"}))
    pipe.close()
    torch.cuda.synchronize()
    for k, g in got:
        ns, sigs, hashed, E = calls[k]
        pk = rsa.assign_public_key(H.RSAPublicKey(H.UnassignedInteger.from_ints(ns, 32, 64), H.Var(H.UnassignedInteger(np.array(E, dtype=np.uint64)))))
        sg = rsa.assign_signature(H.RSASignature(H.UnassignedInteger.from_ints(sigs, 32, 64)))
        res = rsa.verify_pkcs1v15_signature(pk, _hashed_tensor(hashed), sg)
        want = res.emit_advice()
        assert torch.equal(g["st"], res.status) and torch.equal(g["valid"], res.is_valid), k
        assert g["valid"].cpu().tolist()[:3] == [1, 1, 0]
        ok = (res.status == 0)
        assert torch.equal(g["powed"][ok], res.powed.limbs_dev[ok]), k
        assert want.shape == g["img"].shape and torch.equal(g["img"][ok], want[ok]), k
        if k == 1:
            assert int(g["st"][4]) == H.H2R_E_NOT_IN_FIELD and bool((g["img"][4] == 0x5A).all())


def test_record_exports_refuse_a_compact_layout():
    """A witness-only layout (h2r_pow_layout_compact / h2r_verify_layout_compact: off_records = UINT64_MAX) holds no records: every
    export that reads records answers H2R_E_SHAPE instead of adding the sentinel to a pointer (host or device)."""
    from halo2_rsa_amd import _lib
    from halo2_rsa_amd._lib import lib
    L = lib()
    ctx = ctypes.c_void_p()
    p = _lib.H2RParams(64, 2048, 0, -1)
    assert L.h2r_ctx_create(ctypes.byref(p), ctypes.byref(ctx)) == 0
    e = (65537).to_bytes(3, "little")
    full, pl = _lib.H2RPowLayout(), _lib.H2RPowLayout()
    assert L.h2r_pow_fixed_layout(ctx, e, len(e), ctypes.byref(full)) == 0
    assert L.h2r_pow_layout_compact(ctx, ctypes.byref(full), ctypes.byref(pl)) == 0
    vfull, vl = _lib.H2RVerifyLayout(), _lib.H2RVerifyLayout()
    assert L.h2r_verify_layout_fixed(ctx, e, len(e), ctypes.byref(vfull)) == 0
    assert L.h2r_verify_layout_compact(ctx, ctypes.byref(vfull), ctypes.byref(vl)) == 0
    buf = ctypes.create_string_buffer(1 << 16)   # never touched: every call must refuse before it reads or writes
    cfg = _lib.H2RLookupConfig()
    assert L.h2r_lookup_config_default(ctx, 1, ctypes.byref(cfg)) == 0
    S = _lib.H2R_E_SHAPE
    assert L.h2r_pow_trace_flatten(ctx, ctypes.byref(pl), buf, buf) == S
    assert L.h2r_pow_trace_flatten_ex(ctx, ctypes.byref(pl), buf, 0, buf) == S
    assert L.h2r_verify_trace_flatten(ctx, ctypes.byref(vl), buf, buf) == S
    assert L.h2r_pow_trace_emit_stream(ctx, ctypes.byref(pl), buf, 0, 1, 0, buf, 1 << 30, 0, None) == S
    assert L.h2r_pow_trace_check(ctx, ctypes.byref(pl), buf, buf, e, len(e), 0, buf, 0, buf, 1, None, buf, None, None) == S
    assert L.h2r_lookup_hist_verify(ctx, ctypes.byref(cfg), ctypes.byref(vl), buf, 1, None, buf, None) == S
    # the record-read image (no H2R_ADVICE_DIRECT) of a witness-only layout
    assert L.h2r_pow_trace_emit_advice(ctx, ctypes.byref(pl), buf, 0, buf, 0, buf, 1, None, buf, 1 << 30, None) == S
    # the full layouts pass these guards (a host-only ctx then answers H2R_E_UNSUPPORTED where a device is needed)
    assert L.h2r_pow_trace_emit_stream(ctx, ctypes.byref(full), buf, 0, 1, 0, buf, 1 << 30, 0, None) == _lib.H2R_E_UNSUPPORTED
    L.h2r_ctx_destroy(ctx)
