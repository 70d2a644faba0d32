"""The witness in the PROVER'S representation (h2r_advice_repr, include/h2r.h): planar advice columns (H2R_ADVICE_COLUMNS) of
Montgomery-form field elements (H2R_ADVICE_MONTGOMERY).  The reference hands every value to halo2 as `Value<F>`
(big_integer/chip.rs:408, 590, 598; benches/bench.rs:35, 321-329): [3P] F is four 64-bit words holding x * R mod p and halo2 keeps
one contiguous vector per advice column.  The default (row-major, canonical) image is pinned cell for cell against the Python
restatement run on the oracle's stream (tests/test_gpu_parity.py, tests/test_cells_direct.py); here every other representation must
be exactly that image transposed and multiplied by R (tests/advice_ref.py image_to_repr), for every emitter: the record-reading
kernel, the direct cells kernel, the row programs (assert_in_field, the encoded-message check, the hashed-message limbs), the
variable-exponent rows, the layout permutation, the fixed rows and the lookup table."""
import ctypes
import os
import random
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))

R256 = 1 << 256
MODES = [dict(columns=True), dict(montgomery=True), dict(columns=True, montgomery=True)]


def _words(v):
    return (ctypes.c_uint64 * 4)(*[(v >> (64 * k)) & 0xFFFFFFFFFFFFFFFF for k in range(4)])


def _int(arr):
    return sum(int(arr[k]) << (64 * k) for k in range(4))


# ---- host: the conversion itself, the selectors, the table, the ctx's validation (no GPU) ----------------------------------------
@pytest.mark.parametrize("field", ["bn254_fr", "bn254_fq", "pasta_fp", "pasta_fq"])
def test_short_montgomery_product_matches_big_integers(field):
    """h2r_field_eval op 6 = the kernels' short product (17 K multiply-adds for a K-digit value) against x * 2^256 mod p, for every
    digit count; op 8 the generic product; op 7 the way back."""
    import pyref as R
    from halo2_rsa_amd import _lib
    from halo2_rsa_amd._lib import lib
    P = R.FIELD_MODULI[field]
    ctx = ctypes.c_void_p()
    p = _lib.H2RParams(64, 2048, _lib.FIELDS[field], -1)
    assert lib().h2r_ctx_create(ctypes.byref(p), ctypes.byref(ctx)) == 0
    rng = random.Random(7)
    vals = [0, 1, 2, 255, P - 1, P - 2, (1 << 64) - 1, 1 << 64, (1 << 133) - 1, (1 << 160) - 1, 1 << 160]
    for bits in range(1, 255):
        vals.append(rng.getrandbits(bits) | (1 << (bits - 1)))
    out = (ctypes.c_uint64 * 4)()
    for v in vals:
        v %= P
        for op in (6, 8, 9):
            assert lib().h2r_field_eval(ctx, op, _words(v), None, out) == 0
            assert _int(out) == v * R256 % P, (op, hex(v))
        m = _words(v * R256 % P)
        assert lib().h2r_field_eval(ctx, 7, m, None, out) == 0 and _int(out) == v
    lib().h2r_ctx_destroy(ctx)


def test_ctx_repr_validation_and_query():
    from halo2_rsa_amd import _lib
    from halo2_rsa_amd._lib import lib
    assert lib().h2r_abi_version() == _lib.H2R_VERSION
    p = _lib.H2RParams(64, 2048, 0, -1)
    ctx = ctypes.c_void_p()
    sz = ctypes.sizeof(_lib.H2RAdviceRepr)
    bad = [_lib.H2RAdviceRepr(sz - 4, 0, 0), _lib.H2RAdviceRepr(sz, 0x1000, 0), _lib.H2RAdviceRepr(sz, _lib.H2R_ADVICE_COLUMNS, 40),
           _lib.H2RAdviceRepr(sz, _lib.H2R_ADVICE_MONTGOMERY, 4096)]
    want = [_lib.H2R_E_UNSUPPORTED, _lib.H2R_E_SHAPE, _lib.H2R_E_SHAPE, _lib.H2R_E_SHAPE]
    for rp, rc in zip(bad, want):
        assert lib().h2r_ctx_create_ex(ctypes.byref(p), ctypes.byref(rp), ctypes.byref(ctx)) == rc and not ctx.value
    rp = _lib.H2RAdviceRepr(sz, _lib.H2R_ADVICE_COLUMNS | _lib.H2R_ADVICE_MONTGOMERY, 1 << 22)
    assert lib().h2r_ctx_create_ex(ctypes.byref(p), ctypes.byref(rp), ctypes.byref(ctx)) == 0
    got = _lib.H2RAdviceRepr()
    assert lib().h2r_ctx_advice_repr(ctx, ctypes.byref(got)) == 0
    assert (got.struct_size, got.flags, got.col_stride) == (sz, rp.flags, 1 << 22)
    lib().h2r_ctx_destroy(ctx)
    assert lib().h2r_ctx_create_ex(ctypes.byref(p), None, ctypes.byref(ctx)) == 0
    assert lib().h2r_ctx_advice_repr(ctx, ctypes.byref(got)) == 0 and got.flags == 0 and got.col_stride == 0
    lib().h2r_ctx_destroy(ctx)


@pytest.mark.parametrize("field", ["bn254_fr", "pasta_fq"])
def test_fixed_rows_and_table_follow_the_representation(field):
    """Selectors and table entries are field elements like the cells: a Montgomery ctx returns them times R."""
    import pyref as R
    from halo2_rsa_amd import _lib
    from halo2_rsa_amd._lib import lib
    P = R.FIELD_MODULI[field]
    p = _lib.H2RParams(64, 2048, _lib.FIELDS[field], -1)
    plain, mont = ctypes.c_void_p(), ctypes.c_void_p()
    assert lib().h2r_ctx_create(ctypes.byref(p), ctypes.byref(plain)) == 0
    rp = _lib.H2RAdviceRepr(ctypes.sizeof(_lib.H2RAdviceRepr), _lib.H2R_ADVICE_MONTGOMERY, 0)
    assert lib().h2r_ctx_create_ex(ctypes.byref(p), ctypes.byref(rp), ctypes.byref(mont)) == 0
    cfg = _lib.H2RLookupConfig()
    assert lib().h2r_lookup_config_default(plain, 1, ctypes.byref(cfg)) == 0
    seen = 0
    for kind in range(256):
        fa, fb = _lib.H2RFixedRow(), _lib.H2RFixedRow()
        ra = lib().h2r_advice_fixed_row(plain, ctypes.byref(cfg), kind, ctypes.byref(fa))
        rb = lib().h2r_advice_fixed_row(mont, ctypes.byref(cfg), kind, ctypes.byref(fb))
        assert ra == rb
        if ra:
            continue
        seen += 1
        da, db = fa.as_dict(), fb.as_dict()
        for nm in _lib.H2RFixedRow.NAMES:
            assert db[nm] == da[nm] * R256 % P, (kind, nm)
        assert (da["tag_composition"], da["tag_overflow"]) == (db["tag_composition"], db["tag_overflow"])   # tags are fixed-column integers
    assert seen > 40
    n = cfg.n_rows
    ta, va, tb, vb = (np.zeros((n, 4), dtype=np.uint64) for _ in range(4))
    assert lib().h2r_lookup_table_image(plain, ctypes.byref(cfg), ta.ctypes.data, va.ctypes.data) == 0
    assert lib().h2r_lookup_table_image(mont, ctypes.byref(cfg), tb.ctypes.data, vb.ctypes.data) == 0
    for r in range(n):
        assert _int(tb[r]) == _int(ta[r]) * R256 % P and _int(vb[r]) == _int(va[r]) * R256 % P
    lib().h2r_ctx_destroy(plain)
    lib().h2r_ctx_destroy(mont)


# ---- GPU: every emitter, every representation ------------------------------------------------------------------------------------
torch = pytest.importorskip("torch")


@pytest.fixture(scope="module")
def H():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    import halo2_rsa_amd as H_
    return H_


def rand_modulus(rng, bits, odd=True):
    n = rng.getrandbits(bits) | (1 << (bits - 1))
    return n | 1 if odd else n & ~1


def _expect(want_rowmajor, rows, P, mode, col_stride=0, fill=0):
    import advice_ref as AR
    return np.stack([AR.image_to_repr(want_rowmajor[e], rows, P, col_stride=col_stride, fill=fill, **mode) for e in range(want_rowmajor.shape[0])])


def _first_diff(got, want):
    bad = np.argwhere(got != want)
    return None if not len(bad) else "elem %d byte %d" % (int(bad[0][0]), int(bad[0][1]))


SHAPES = [(64, 32, "bn254_fr"), (64, 16, "bn254_fq"), (32, 128, "pasta_fp"), (64, 12, "bn254_fq"), (64, 48, "pasta_fq"),
          (32, 8, "bn254_fr"), (64, 64, "bn254_fr"), (64, 4, "pasta_fq"), (32, 96, "pasta_fq"), (64, 24, "bn254_fr")]


@pytest.mark.gpu
@pytest.mark.parametrize("w,L,field", SHAPES)
def test_mul_mod_and_pow_images_in_every_representation(H, w, L, field):
    """The ten shapes / four fields of test_direct_image_equals_record_image: mul_mod batches (even / all-ones / zero operands among them)
    and a pow call, through the record-reading kernel AND the direct cells kernel, in the three non-default representations."""
    import pyref as R
    from halo2_rsa_amd._lib import lib
    P = R.FIELD_MODULI[field]
    rng = random.Random(w * 1000 + L)
    batch = 3 if L >= 64 else 5
    N = [rand_modulus(rng, w * L, odd=(i != 1)) for i in range(batch)]
    A = [rng.randrange(n) for n in N]
    B = [rng.randrange(n) for n in N]
    A[2] = B[2] = N[2] - 1
    A[0] = 0
    base = H.BigIntChip(w, w * L, field=field)
    rows = int(lib().h2r_advice_rows(base._ctx))
    res0 = base.mul_mod(base.assign_integer(A), base.assign_integer(B), base.assign_integer(N))
    want_mm = res0.emit_advice().cpu().numpy()
    e = 0b1011
    p0 = base.pow_mod_fixed_exp(base.assign_integer(A), e, base.assign_integer(N))
    want_pow = p0.emit_advice().cpu().numpy()
    prows = want_pow.shape[1] // 160
    for mode in MODES:
        chip = H.BigIntChip(w, w * L, field=field, **mode)
        res = chip.mul_mod(chip.assign_integer(A), chip.assign_integer(B), chip.assign_integer(N))
        exp = _expect(want_mm, rows, P, mode)
        for direct in (False, True):
            got = res.emit_advice(direct=direct).cpu().numpy()
            assert _first_diff(got, exp) is None, (mode, direct, "mul_mod", _first_diff(got, exp))
        pres = chip.pow_mod_fixed_exp(chip.assign_integer(A), e, chip.assign_integer(N))
        exp = _expect(want_pow, prows, P, mode)
        for direct in (False, True):
            got = pres.emit_advice(direct=direct).cpu().numpy()
            assert _first_diff(got, exp) is None, (mode, direct, "pow", _first_diff(got, exp))


@pytest.mark.gpu
@pytest.mark.parametrize("w,L,field", [(64, 32, "bn254_fr"), (32, 16, "pasta_fq"), (64, 12, "bn254_fq"), (64, 64, "bn254_fr"), (32, 96, "pasta_fp")])
def test_inconsistent_mul_mod_in_every_representation(H, w, L, field):
    """Records whose q, r are NOT the quotient and remainder (R plane bumped by one; a second element with its Q plane bumped): the
    direct kernel leaves its fast rows for the general path (d = x - y != 0, the inverse witness, eq_bit 0, the carries of a
    non-zero difference); in a Montgomery ctx that path works on cells, so its image must still be the canonical ctx's image of the
    SAME records, transposed and multiplied by R (the canonical one is checked against the gate in tests/test_cells_direct.py)."""
    import pyref as R
    from halo2_rsa_amd import _lib
    from halo2_rsa_amd._lib import lib
    P = R.FIELD_MODULI[field]
    rng = random.Random(5 * w + L)
    N = [rand_modulus(rng, w * L) for _ in range(4)]
    A = [rng.randrange(n) for n in N]
    B = [rng.randrange(n) for n in N]
    P_IDX = {nm: k for k, nm in enumerate(_lib.PLANES)}

    def run(kw):
        chip = H.BigIntChip(w, w * L, field=field, **kw)
        res = chip.mul_mod(chip.assign_integer(A), chip.assign_integer(B), chip.assign_integer(N))
        torch.cuda.synchronize()
        lo = chip.layout
        for elem, plane, limb in ((1, "R", 0), (3, "Q", 1)):
            off = elem * lo.record_stride + lo.plane_off[P_IDX[plane]] + limb * lo.limb_bytes
            v = int.from_bytes(res.trace.buf[off:off + lo.limb_bytes].cpu().numpy().tobytes(), "little")
            v = v + 1 if v + 1 < (1 << w) else v - 1
            res.trace.buf[off:off + lo.limb_bytes] = torch.from_numpy(np.frombuffer(v.to_bytes(lo.limb_bytes, "little"), dtype=np.uint8).copy()).cuda()
        rows = int(lib().h2r_advice_rows(chip._ctx))
        return rows, res.emit_advice(direct=True).cpu().numpy()

    rows, want = run({})
    for mode in MODES:
        r2, got = run(mode)
        assert r2 == rows
        exp = _expect(want, rows, P, mode)
        assert _first_diff(got, exp) is None, (mode, _first_diff(got, exp))


@pytest.mark.gpu
@pytest.mark.parametrize("arrangement", ["element_major", "column_major"])
def test_columns_of_a_fixed_stride_and_guard_bytes(H, arrangement):
    """halo2's columns have 2^k rows: col_stride = 2^k * 32, the region starts at some row r0 of the caller's columns.  Both
    [element][column][row] and [column][element][row]; nothing outside the region's rows is written (0xA5 guard)."""
    import pyref as R
    from halo2_rsa_amd import _lib
    from halo2_rsa_amd._lib import lib
    P = R.FIELD_MODULI["bn254_fr"]
    rng = random.Random(99)
    batch, e = 3, 17
    N = [rand_modulus(rng, 2048) for _ in range(batch)]
    X = [rng.randrange(n) for n in N]
    base = H.BigIntChip(64, 2048)
    want = base.pow_mod_fixed_exp(base.assign_integer(X), e, base.assign_integer(N)).emit_advice().cpu().numpy()
    rows = want.shape[1] // 160
    k_rows = 1 << 15                       # 2^15-row columns hold the 2 + 6 * 3,974 rows of e = 17
    assert rows + 5 <= k_rows
    for r0 in (0, 1, 2, 3, 5):             # every 128-byte phase of the first row
        if arrangement == "element_major":
            cs, es = k_rows * 32, 5 * k_rows * 32
        else:
            es, cs = k_rows * 32, batch * k_rows * 32
        chip = H.BigIntChip(64, 2048, columns=True, montgomery=True, col_stride=cs)
        pres = chip.pow_mod_fixed_exp(chip.assign_integer(X), e, chip.assign_integer(N))
        total = 5 * batch * k_rows * 32
        buf = torch.full((total + 256,), 0xA5, dtype=torch.uint8, device="cuda")
        base_off = (-buf.data_ptr()) % 256
        n_dev = pres.inputs[3]
        rc = lib().h2r_pow_trace_emit_advice(chip._ctx, ctypes.byref(pres.trace.pow_layout), n_dev.data_ptr(), _lib.H2R_ADVICE_DIRECT, None, 0,
                                             pres.workspace.data_ptr(), batch, pres.status.data_ptr(), buf.data_ptr() + base_off + r0 * 32, es,
                                             chip._stream())
        assert rc == 0
        torch.cuda.synchronize()
        host = buf.cpu().numpy()[base_off:base_off + total]
        exp = np.full(total, 0xA5, dtype=np.uint8)
        for el in range(batch):
            img = __import__("advice_ref").image_to_repr(want[el], rows, P, columns=True, montgomery=True).reshape(5, rows * 32)
            for c in range(5):
                o = el * es + c * cs + r0 * 32
                exp[o:o + rows * 32] = img[c]
        assert np.array_equal(host, exp), (arrangement, r0, int(np.argwhere(host != exp)[0][0]))
    # a stride that cannot hold the rows, and an element stride that overlaps the columns, are refused
    small = H.BigIntChip(64, 2048, columns=True, col_stride=1024)
    ps = small.pow_mod_fixed_exp(small.assign_integer(X), e, small.assign_integer(N))
    assert lib().h2r_pow_trace_emit_advice(small._ctx, ctypes.byref(ps.trace.pow_layout), ps.inputs[3].data_ptr(), _lib.H2R_ADVICE_DIRECT, None, 0,
                                           ps.workspace.data_ptr(), batch, ps.status.data_ptr(), buf.data_ptr(), 5 * 1024, small._stream()) == _lib.H2R_E_SHAPE


@pytest.mark.gpu
@pytest.mark.parametrize("mode", MODES)
def test_whole_elements_in_every_representation(H, mode):
    """modpow_public_key (in-field rows + pow rows), the whole verify element (seed, assert_in_field, pow, encoded-message check) for
    KAT1 / KAT2 / BAD, and the Var arm (to_bits, select rows): the row programs and var_rows_kernel in the consumer's representation."""
    import json
    import pyref as R
    from halo2_rsa_amd._lib import lib
    P = R.FIELD_MODULI["bn254_fr"]
    with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "halo2_rsa_golden.json")) as f:
        kats = json.load(f)["rsa_kats"]
    ns, sigs = [int(k["n"]) for k in kats], [int(k["sig"]) for k in kats]
    hashes = [int(k["hash"], 16) if isinstance(k["hash"], str) else int(k["hash"]) for k in kats] if "hash" in kats[0] else None

    def run(repr_kw):
        rsa = H.RSAChip(2048, 5, **repr_kw)
        chip = rsa.bigint_chip()
        pk = rsa.assign_public_key(H.RSAPublicKey(H.UnassignedInteger.from_ints(ns, 32, 64), H.Fix(65537)))
        x = chip.assign_integer(sigs)
        res = rsa.modpow_public_key(x, pk)
        out = {"modpow": res.emit_modpow_advice().cpu().numpy(), "modpow_direct": res.emit_modpow_advice(direct=True).cpu().numpy()}
        # the Var arm: 5-bit exponents
        pkv = rsa.assign_public_key(H.RSAPublicKey(H.UnassignedInteger.from_ints(ns, 32, 64), H.Var(H.UnassignedInteger.from_ints([19, 31, 1], 1, 64))))
        rv = rsa.modpow_public_key(x, pkv)
        out["var"] = rv.emit_advice().cpu().numpy()
        out["var_direct"] = rv.emit_advice(direct=True).cpu().numpy()
        return out

    want = run({})
    got = run(mode)
    for key in want:
        rows = want[key].shape[1] // 160
        exp = _expect(want[key], rows, P, mode)
        assert _first_diff(got[key], exp) is None, (mode, key, _first_diff(got[key], exp))


@pytest.mark.gpu
@pytest.mark.parametrize("mode", MODES)
def test_verify_element_and_layout_permutation(H, mode):
    import json
    import pyref as R
    from halo2_rsa_amd import _lib
    from halo2_rsa_amd._lib import lib
    P = R.FIELD_MODULI["bn254_fr"]
    with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "halo2_rsa_golden.json")) as f:
        kats = json.load(f)["rsa_kats"]
    ns, sigs = [int(k["n"]) for k in kats], [int(k["sig"]) for k in kats]
    import hashlib
    h = int.from_bytes(hashlib.sha256(b"hello world").digest(), "big")
    hashed = [[(h >> (64 * j)) & 0xFFFFFFFFFFFFFFFF for j in range(4)]] * 3

    def run(repr_kw):
        rsa = H.RSAChip(2048, 5, **repr_kw)
        chip = rsa.bigint_chip()
        pk = rsa.assign_public_key(H.RSAPublicKey(H.UnassignedInteger.from_ints(ns, 32, 64), H.Fix(65537)))
        sig = rsa.assign_signature(H.RSASignature(H.UnassignedInteger.from_ints(sigs, 32, 64)))
        hd = torch.tensor(np.array(hashed, dtype=np.uint64).view(np.int64), device="cuda")
        res = rsa.verify_pkcs1v15_signature(pk, hd, sig)
        img = res.emit_advice()
        kinds = res.row_kinds()
        # a custom layout: swap the pair columns of the mul_add rows and of the select rows
        lay = _lib.H2RAdviceLayout()
        ks = (ctypes.c_uint8 * 2)(6, 7)                      # MUL_ADD, ADD
        cols = ((ctypes.c_uint8 * 5) * 2)((1, 0, 2, 3, 4), (1, 0, 2, 3, 4))
        assert lib().h2r_advice_layout_custom(chip._ctx, ks, cols, 2, ctypes.byref(lay)) == 0
        kd = torch.tensor(kinds, device="cuda")
        permuted = img.clone()
        assert lib().h2r_advice_apply_layout(chip._ctx, ctypes.byref(lay), kd.data_ptr(), len(kinds), permuted.data_ptr(), permuted.shape[1], 3,
                                             res.status.data_ptr(), chip._stream()) == 0
        torch.cuda.synchronize()
        return img.cpu().numpy(), permuted.cpu().numpy(), res.is_valid.cpu().tolist()

    w_img, w_perm, w_valid = run({})
    g_img, g_perm, g_valid = run(mode)
    assert w_valid == g_valid == [1, 1, 0]
    rows = w_img.shape[1] // 160
    for nm, w_, g_ in (("image", w_img, g_img), ("permuted", w_perm, g_perm)):
        exp = _expect(w_, rows, P, mode)
        assert _first_diff(g_, exp) is None, (mode, nm, _first_diff(g_, exp))
    assert not np.array_equal(w_img, w_perm)


@pytest.mark.gpu
def test_lookup_columns_in_montgomery_form(H):
    """h2r_lookup_permuted_columns of a Montgomery ctx: theta comes in times R, A' / S' go out times R, the ORDER is that of the
    canonical integers (the field's Ord) either way."""
    import pyref as R
    P = R.FIELD_MODULI["bn254_fr"]
    w, L = 64, 8
    rng = random.Random(77)
    n = rng.getrandbits(w * L) | (1 << (w * L - 1)) | 1
    x = rng.randrange(n)
    usable = (1 << 12) - 6
    thetas = [rng.randrange(P), P - 1]
    outs = []
    for mont in (False, True):
        chip = H.BigIntChip(w, w * L, montgomery=mont)
        la = H.LookupArgument(chip, rsa_chip=False)
        x_dev, n_dev = chip.assign_integer([x, x]), chip.assign_integer([n, n])
        res = chip.pow_mod_fixed_exp(x_dev, 3, n_dev)
        hist = la.new_hist(2)
        la.hist_values(x_dev.limbs_dev, w, 8, hist)
        la.hist_records(res.trace, hist, res.status)
        th = [t * R256 % P for t in thetas] if mont else thetas
        a_perm, s_perm, status = la.permuted_columns(hist, th, usable)
        torch.cuda.synchronize()
        assert status.cpu().tolist() == [0, 0]
        outs.append((a_perm.cpu().numpy(), s_perm.cpu().numpy()))
    for col in range(2):
        plain = outs[0][col].reshape(-1, 4).view("<u8") if False else outs[0][col].reshape(-1, 32)
        mont = outs[1][col].reshape(-1, 32)
        # distinct values are few (runs): convert each distinct 32-byte value once
        uniq, inv = np.unique(plain, axis=0, return_inverse=True)
        conv = np.stack([np.frombuffer((int.from_bytes(u.tobytes(), "little") * R256 % P).to_bytes(32, "little"), dtype=np.uint8) for u in uniq])
        assert np.array_equal(conv[inv.reshape(-1)], mont), "A'" if col == 0 else "S'"


@pytest.mark.gpu
@pytest.mark.parametrize("mode", MODES)
def test_fresh_ops_and_the_verifier_region_in_every_representation(H, mode):
    """The stand-alone Fresh-integer ops (row programs with is_zero inverse witnesses: full-size field elements) and the whole
    RSASignatureVerifier region -- hashed-message limb rows in front of the verify element, TWO calls writing ONE image, which needs
    columns of a fixed stride -- in the consumer's representation."""
    import pyref as R
    P = R.FIELD_MODULI["bn254_fr"]
    rng = random.Random(5150)
    bits, batch = 1024, 4
    N = [rand_modulus(rng, bits) for _ in range(batch)]
    A = [rng.randrange(n) for n in N]
    B = [rng.randrange(n) for n in N]
    B[1] = A[1]                                  # an equal pair: is_equal's d = 0 branch next to the d != 0 ones
    k_rows = 1 << 13

    def fresh(kw):
        chip = H.BigIntChip(64, bits, **kw)
        a, b, n = chip.assign_integer(A), chip.assign_integer(B), chip.assign_integer(N)
        out = {}
        for name, res in (("add", chip.add(a, b)), ("sub_mod", chip.sub_mod(a, b, n)), ("is_equal_fresh", chip.is_equal_fresh(a, b)),
                          ("is_less_than", chip.is_less_than(a, b)), ("is_zero", chip.is_zero(a))):
            out[name] = res.emit_advice().cpu().numpy()
        return out

    want, got = fresh({}), fresh(mode)
    for name in want:
        rows = want[name].shape[1] // 160
        exp = _expect(want[name], rows, P, mode)
        assert _first_diff(got[name], exp) is None, (mode, name, _first_diff(got[name], exp))

    # the verifier's region from message bytes
    import json
    with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "halo2_rsa_golden.json")) as f:
        kats = json.load(f)["rsa_kats"]
    ns, sigs = [int(k["n"]) for k in kats], [int(k["sig"]) for k in kats]

    def region(kw):
        rsa = H.RSAChip(2048, 5, **kw)
        ver = H.RSASignatureVerifier(rsa)
        pk = H.RSAPublicKey(H.UnassignedInteger.from_ints(ns, 32, 64), H.Fix(65537))
        res = ver.verify_pkcs1v15_signature(pk, b"hello world", H.RSASignature(H.UnassignedInteger.from_ints(sigs, 32, 64)))
        return res.emit_advice(with_hashed_msg=True).cpu().numpy(), res.is_valid.cpu().tolist()

    w_img, w_valid = region({})
    rows = w_img.shape[1] // 160
    kw = dict(mode)
    if kw.get("columns"):
        kw["col_stride"] = ((rows * 32 + 4095) // 4096) * 4096
    g_img, g_valid = region(kw)
    assert w_valid == g_valid == [1, 1, 0]
    exp = _expect(w_img, rows, P, mode, col_stride=kw.get("col_stride", 0))
    if kw.get("columns"):           # (the columns are longer than the region: compare what the image covers)
        cs = kw["col_stride"]
        g5 = g_img.reshape(3, 5, cs)[:, :, :rows * 32]
        e5 = exp.reshape(3, -1)
        e5 = np.stack([np.stack([e5[e, c * cs:c * cs + rows * 32] for c in range(5)]) for e in range(3)])
        assert np.array_equal(g5, e5)
    else:
        assert _first_diff(g_img, exp) is None, (mode, "verifier region", _first_diff(g_img, exp))
    # without a fixed stride two calls cannot share packed columns: refused, not mis-written
    if mode.get("columns"):
        rsa = H.RSAChip(2048, 5, **mode)
        res = H.RSASignatureVerifier(rsa).verify_pkcs1v15_signature(H.RSAPublicKey(H.UnassignedInteger.from_ints(ns, 32, 64), H.Fix(65537)), b"hello world",
                                                                    H.RSASignature(H.UnassignedInteger.from_ints(sigs, 32, 64)))
        with pytest.raises(ValueError):
            res.emit_advice(with_hashed_msg=True)
