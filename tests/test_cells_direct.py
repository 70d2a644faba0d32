"""cells_kernel (H2R_ADVICE_DIRECT): the 5-column advice image written DIRECTLY from a mul_mod's operands -- what the reference does
when it puts every value straight into main-gate cells (main_gate.mul_add big_integer/chip.rs:408, range_chip.assign :590, :598,
:880-885, is_equal_muled :851-893).  The record-reading form (advice_kernel) is pinned cell for cell against the Python restatement
run on the ORACLE's stream (tests/test_gpu_parity.py::test_advice_image); here the direct form must equal it byte for byte --
every supported limb shape, mul_mod batches, pow traces, calls that wrote no records at all, whole modpow_public_key / verify
elements, BASELINE config 2 at full size -- and, on a mul_mod whose q, r are NOT its quotient and remainder (never produced by the
library: the general path with main_gate.is_zero's inverse witnesses), every row must still satisfy the main-gate equation."""
import ctypes
import os
import random
import sys

import numpy as np
import pytest

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def H():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    import halo2_rsa_amd as H_
    return H_


def rand_modulus(rng, bits, odd=True):
    n = rng.getrandbits(bits) | (1 << (bits - 1))
    return n | 1 if odd else n & ~1


def _first_diff(got, want, rows, kinds):
    g, w = got.reshape(-1, rows, 160), want.reshape(-1, rows, 160)
    bad = np.argwhere(g != w)
    if not len(bad):
        return None
    e, r, b = (int(v) for v in bad[0])
    return "elem %d row %d (kind %d) cell %d" % (e, r, int(kinds[r]), b // 32)


@pytest.mark.parametrize("w,L,field", [(64, 32, "bn254_fr"), (64, 16, "bn254_fq"), (32, 128, "pasta_fp"), (64, 12, "bn254_fq"), (64, 48, "pasta_fq"),
                                       (32, 8, "bn254_fr"), (64, 64, "bn254_fr"), (64, 4, "pasta_fq"), (32, 96, "pasta_fq"), (64, 24, "bn254_fr")])
def test_direct_image_equals_record_image(H, w, L, field):
    """mul_mod batches (even / all-ones / zero operands among them), a short and the RSA exponent, and pow calls that wrote no records:
    the direct image is the record image, byte for byte; the two constant rows of a fixed-exponent pow element included."""
    from halo2_rsa_amd._lib import lib
    chip = H.BigIntChip(w, w * L, field=field)
    rng = random.Random(w * 1000 + L)
    batch = 5
    N = [rand_modulus(rng, w * L, odd=(i != 1)) for i in range(batch)]
    A = [rng.randrange(n) for n in N]
    B = [rng.randrange(n) for n in N]
    A[2] = B[2] = N[2] - 1
    A[3] = 0
    rows = int(lib().h2r_advice_rows(chip._ctx))
    kinds = np.zeros(rows, dtype=np.uint8)
    assert lib().h2r_advice_row_kinds(chip._ctx, kinds.ctypes.data) == 0
    res = chip.mul_mod(chip.assign_integer(A), chip.assign_integer(B), chip.assign_integer(N))
    want = res.emit_advice().cpu().numpy()
    got = res.emit_advice(direct=True).cpu().numpy()
    assert _first_diff(got, want, rows, kinds) is None, _first_diff(got, want, rows, kinds)
    for e in (0b1011, 65537):
        pres = chip.pow_mod_fixed_exp(chip.assign_integer(A), e, chip.assign_integer(N))
        want = pres.emit_advice().cpu().numpy()
        got = pres.emit_advice(direct=True).cpu().numpy()
        assert np.array_equal(got[:, :320], want[:, :320])
        assert _first_diff(got[:, 320:], want[:, 320:], rows, kinds) is None, (e, _first_diff(got[:, 320:], want[:, 320:], rows, kinds))
        T = pres.trace.num_mul_mods
        bare = chip.pow_mod_fixed_exp(chip.assign_integer(A), e, chip.assign_integer(N), want_trace=False,
                                      workspace=torch.empty(chip.workspace_bytes(batch, T), dtype=torch.uint8, device="cuda"))
        assert bare.trace is None
        assert np.array_equal(bare.emit_advice(direct=True).cpu().numpy(), want), ("no-record call", e)
        with pytest.raises(ValueError):
            bare.emit_advice()


def test_direct_image_into_unaligned_rows(H):
    """The kernel cuts an item's first chunk so that later ones start on 128-byte lines: every start alignment (row offsets 0..3 of
    the element, i.e. out_stride not a multiple of 128) gives the same bytes."""
    from halo2_rsa_amd._lib import lib
    from halo2_rsa_amd import _lib
    chip = H.BigIntChip(64, 2048)
    rng = random.Random(5)
    N = [rand_modulus(rng, 2048) for _ in range(3)]
    X = [rng.randrange(n) for n in N]
    pres = chip.pow_mod_fixed_exp(chip.assign_integer(X), 17, chip.assign_integer(N))
    want = pres.emit_advice().cpu().numpy()
    nbytes = want.shape[1]
    n_dev = pres.inputs[3]
    for extra_rows in (1, 2, 3):
        stride = nbytes + extra_rows * 160
        buf = torch.full((3 * stride + 4096,), 0xA5, dtype=torch.uint8, device="cuda")
        base = (-buf.data_ptr()) % 256   # a 256-byte aligned start inside the buffer
        assert lib().h2r_pow_trace_emit_advice(chip._ctx, ctypes.byref(pres.trace.pow_layout), n_dev.data_ptr(), _lib.H2R_ADVICE_DIRECT, None, 0,
                                               pres.workspace.data_ptr(), 3, pres.status.data_ptr(), buf.data_ptr() + base, stride, chip._stream()) == 0
        torch.cuda.synchronize()
        host = buf.cpu().numpy()
        for e in range(3):
            assert np.array_equal(host[base + e * stride:base + e * stride + nbytes], want[e]), (extra_rows, e)
            assert (host[base + e * stride + nbytes:base + (e + 1) * stride] == 0xA5).all(), "bytes behind an element's image were written"


@pytest.mark.parametrize("w,L,field", [(64, 32, "bn254_fr"), (32, 16, "pasta_fq")])
def test_direct_image_of_an_inconsistent_mul_mod_satisfies_the_gate(H, w, L, field):
    """q, r that are NOT the quotient and remainder of a * b (the record's R plane bumped by one): the kernel's general path --
    d = x - y != 0 in is_equal, its inverse witness, eq_bit falling to 0 -- must still give a satisfying assignment of every row
    of is_equal_muled (the main-gate equation with the fixed row of the row's kind), the final eq_bit must be 0, and the closing assert_one row
    must hold that 0."""
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
    import pyref as R
    import advice_ref as AR
    from halo2_rsa_amd import _lib
    from halo2_rsa_amd._lib import lib
    chip = H.BigIntChip(w, w * L, field=field)
    P = R.FIELD_MODULI[field]
    rng = random.Random(11 * w + L)
    n = rand_modulus(rng, w * L)
    a, b = rng.randrange(n), rng.randrange(n)
    res = chip.mul_mod(chip.assign_integer([a, a]), chip.assign_integer([b, b]), chip.assign_integer([n, n]))
    torch.cuda.synchronize()
    lo = chip.layout
    P_IDX = {nm: k for k, nm in enumerate(_lib.PLANES)}
    off = lo.record_stride + lo.plane_off[P_IDX["R"]]             # element 1's r limb 0: + 1 (r < n - 1 with overwhelming probability)
    v = int.from_bytes(res.trace.buf[off:off + lo.limb_bytes].cpu().numpy().tobytes(), "little")
    assert v + 1 < (1 << w)
    res.trace.buf[off:off + lo.limb_bytes] = torch.from_numpy(np.frombuffer((v + 1).to_bytes(lo.limb_bytes, "little"), dtype=np.uint8).copy()).cuda()
    rows = int(lib().h2r_advice_rows(chip._ctx))
    kinds = np.zeros(rows, dtype=np.uint8)
    assert lib().h2r_advice_row_kinds(chip._ctx, kinds.ctypes.data) == 0
    img = res.emit_advice(direct=True).cpu().numpy().reshape(2, rows, 160)
    good = res.emit_advice().cpu().numpy().reshape(2, rows, 160)
    assert np.array_equal(img[0], good[0])                        # the untouched element
    la = H.LookupArgument(chip, rsa_chip=(w == 64))
    fixed = {}
    for k in sorted(set(kinds.tolist())):
        fr = _lib.H2RFixedRow()
        assert lib().h2r_advice_fixed_row(chip._ctx, ctypes.byref(la.cfg), k, ctypes.byref(fr)) == 0
        fixed[k] = fr.as_dict()
    cells = [[int.from_bytes(img[1, r, 32 * c:32 * c + 32].tobytes(), "little") for c in range(5)] for r in range(rows)]
    n_inv = 0
    for r in range(rows - 1):                                      # every row but the record's last one ...
        f = fixed[int(kinds[r])]
        assert AR.gate_residual(cells[r], cells[r + 1][4] if r + 1 < rows else 0, f, P) == 0, (r, int(kinds[r]))
        if int(kinds[r]) == AR.ROW_ISZERO_INV and cells[r][0] != 0:
            assert cells[r][0] * cells[r][1] % P == 1 and cells[r][2] == 0
            n_inv += 1
    assert n_inv >= 1, "no is_zero inverse witness in the image of an inconsistent mul_mod"
    assert cells[rows - 2][2] == 0, "final eq_bit of an inconsistent mul_mod"   # the last `and` row: [e1, f2, e2]
    # ... which is assert_equal_muled's main_gate.assert_one(eq_bit) (chip.rs:1062): the ONE row an inconsistent mul_mod cannot satisfy --
    # it holds the eq_bit it was given (0), its gate reads a - 1 = -1.  (The reference's MockProver flags exactly this row.)
    assert int(kinds[rows - 1]) == 17 and cells[rows - 1] == [0, 0, 0, 0, 0]
    assert AR.gate_residual(cells[rows - 1], 0, fixed[17], P) == P - 1


def test_modpow_public_key_element_without_records(H, golden):
    """RSAChip::modpow_public_key as cells only: the call writes no records (chain + assert_in_field witness), and
    h2r_modpow_public_key_emit_advice gives [assert_in_field rows] [pow rows] -- equal to the rows the record-based exports give
    for the same inputs (KAT1 / KAT2 / BAD + random signatures, one of them not in the field: skipped, its bytes untouched)."""
    from halo2_rsa_amd import _lib
    from halo2_rsa_amd._lib import lib
    chip = H.BigIntChip(64, 2048)
    kats = golden["rsa_kats"]
    rng = random.Random(21)
    ns = [int(k["n"]) for k in kats] + [rand_modulus(rng, 2048) for _ in range(3)]
    xs = [int(k["sig"]) for k in kats] + [rng.randrange(n) for n in ns[3:]]
    xs[4] = ns[4] + 5
    full = chip.pow_mod_fixed_exp(chip.assign_integer(xs), 65537, chip.assign_integer(ns), check_in_field=True)
    want_if = full.in_field.emit_advice(chip.assign_integer(xs), chip.assign_integer(ns)).cpu().numpy()
    want_pow = full.emit_advice().cpu().numpy()
    pl = full.trace.pow_layout
    sec = (ctypes.c_uint64 * 2)()
    total = int(lib().h2r_modpow_public_key_advice_rows(chip._ctx, ctypes.byref(pl), sec))
    assert list(sec) == [want_if.shape[1] // 160, want_pow.shape[1] // 160] and total == sum(sec)
    assert list(sec) == [1532, 2 + 19 * 3974]
    bare = chip.pow_mod_fixed_exp(chip.assign_integer(xs), 65537, chip.assign_integer(ns), want_trace=False, check_in_field=True,
                                  workspace=torch.empty(chip.workspace_bytes(6, pl.num_mul_mods), dtype=torch.uint8, device="cuda"))
    assert bare.trace is None and bare.in_field is not None
    out = torch.full((6, total * 160), 0x5A, dtype=torch.uint8, device="cuda")
    img = bare.emit_modpow_advice(out=out).cpu().numpy()
    assert bare.status.cpu().tolist() == [0, 0, 0, 0, H.H2R_E_NOT_IN_FIELD, 0]
    for i in range(6):
        if i == 4:
            assert (img[i] == 0x5A).all()
            continue
        assert np.array_equal(img[i, :sec[0] * 160], want_if[i]), ("in_field", i)
        assert np.array_equal(img[i, sec[0] * 160:], want_pow[i]), ("pow", i)
    # the same export on the call that has records: from the records, and directly
    assert np.array_equal(full.emit_modpow_advice().cpu().numpy()[:4], img[:4])
    assert np.array_equal(full.emit_modpow_advice(direct=True).cpu().numpy()[:4], img[:4])


def test_pipelined_modpow_advice_calls(H, golden):
    """h2r_pipeline_modpow_public_key_advice: five pipelined calls over two buffer sets (different inputs per call, one element not in the
    field, one call with ONE modulus for the whole batch), inputs overwritten right after each call returns -- every image equals the
    image of a plain call on the same inputs; an element with a status keeps its bytes."""
    chip = H.BigIntChip(64, 2048)
    rng = random.Random(0x68327273 + 46)
    B, depth = 24, 2
    pl = chip.pow_fixed_layout(65537)
    sec = (ctypes.c_uint64 * 2)()
    from halo2_rsa_amd._lib import lib
    rows = int(lib().h2r_modpow_public_key_advice_rows(chip._ctx, ctypes.byref(pl), sec))
    ifs = chip.in_field_layout()[0]
    sets = [dict(ws=torch.empty(chip.workspace_bytes(B, pl.num_mul_mods), dtype=torch.uint8, device="cuda"),
                 out=torch.zeros((B, chip.num_limbs), dtype=chip.torch_dtype, device="cuda"), st=torch.zeros(B, dtype=torch.uint8, device="cuda"),
                 inf=torch.zeros(B * ifs, dtype=torch.uint8, device="cuda"), img=torch.empty((B, rows * 160), dtype=torch.uint8, device="cuda"))
            for _ in range(depth)]
    pipe = H.Pipeline(chip, depth, 2)
    calls, got = [], []
    x_stage = n_stage = None
    for k in range(5):
        shared = k == 3
        N = [rand_modulus(rng, 2048) for _ in range(1 if shared else B)]
        X = [rng.randrange(N[0] if shared else N[i]) for i in range(B)]
        if k == 1:
            X[5] = N[5] + 1
        calls.append((X, N))
        x_new, n_new = chip.assign_integer(X), chip.assign_integer(N)
        if x_stage is not None and x_stage.limbs_dev.shape == x_new.limbs_dev.shape and n_stage.limbs_dev.shape == n_new.limbs_dev.shape:
            x_stage.limbs_dev.copy_(x_new.limbs_dev); n_stage.limbs_dev.copy_(n_new.limbs_dev)     # a producer refilling its staging buffers
        else:
            x_stage, n_stage = x_new, n_new
        s = sets[k % depth]
        if k >= depth:
            got.append((k - depth, s["img"].clone(), s["st"].clone()))     # (stream-ordered behind the join the previous call made)
        s["img"].fill_(0x5A)
        pipe.modpow_public_key_advice(x_stage, 65537, n_stage, s["ws"], s["out"], s["st"], s["inf"], s["img"])
    pipe.join()
    for k in range(5 - depth, 5):
        got.append((k, sets[k % depth]["img"].clone(), sets[k % depth]["st"].clone()))
    pipe.close()
    torch.cuda.synchronize()
    assert sorted(g[0] for g in got) == list(range(5))
    for k, img, st in got:
        X, N = calls[k]
        plain = chip.pow_mod_fixed_exp(chip.assign_integer(X), 65537, chip.assign_integer(N), want_trace=False, check_in_field=True,
                                       workspace=torch.empty(chip.workspace_bytes(B, pl.num_mul_mods), dtype=torch.uint8, device="cuda"))
        want = torch.full_like(img, 0x5A)
        plain.emit_modpow_advice(out=want)
        assert torch.equal(st, plain.status), k
        assert torch.equal(img, want), k
        if k == 1:
            assert int(st[5]) == H.H2R_E_NOT_IN_FIELD and bool((img[5] == 0x5A).all())


def test_verify_element_direct(H, golden):
    """h2r_verify_emit_advice with H2R_ADVICE_DIRECT: the pow section written from the operands, the whole element unchanged."""
    rsa = H.RSAChip(2048, 5)
    kats = golden["rsa_kats"]
    ns = [int(k["n"]) for k in kats]
    sigs = [int(k["sig"]) for k in kats]
    hashed = [int(k["hashed"]) for k in kats]
    pk = rsa.assign_public_key(H.RSAPublicKey(H.UnassignedInteger.from_ints(ns, 32, 64), H.Fix(65537)))
    sg = rsa.assign_signature(H.RSASignature(H.UnassignedInteger.from_ints(sigs, 32, 64)))
    res = rsa.verify_pkcs1v15_signature(pk, hashed, sg)
    assert torch.equal(res.emit_advice(direct=True), res.emit_advice())


def test_verify_element_from_two_threads_and_streams(H, golden):
    """h2r_verify_emit_advice forks its short row programs onto a side stream of the ctx and joins them: two host threads calling it on
    ONE ctx at the same time, each on a stream and into an image of its own, both get the image a lone call writes (every round)."""
    import threading
    rsa = H.RSAChip(2048, 5)
    kats = golden["rsa_kats"]
    rng = random.Random(0x68327273 + 44)
    B = 48
    N = [rand_modulus(rng, 2048) for _ in range(B)]
    pk = rsa.assign_public_key(H.RSAPublicKey(H.UnassignedInteger.from_ints(N, 32, 64), H.Fix(65537)))
    sg = rsa.assign_signature(H.RSASignature(H.UnassignedInteger.from_ints([rng.randrange(n) for n in N], 32, 64)))
    res = rsa.verify_pkcs1v15_signature(pk, [rng.getrandbits(256) for _ in range(B)], sg)
    want = res.emit_advice(direct=True)
    assert torch.equal(want, res.emit_advice())
    torch.cuda.synchronize()
    got, errs = {}, []

    def worker(i):
        try:
            st = torch.cuda.Stream()
            for rnd in range(6):
                with torch.cuda.stream(st):
                    img = res.emit_advice(direct=(i + rnd) % 2 == 0)
                st.synchronize()
                if not torch.equal(img, want):
                    errs.append((i, rnd))
                got[i] = rnd
        except Exception as ex:   # noqa: BLE001 -- reported below
            errs.append((i, repr(ex)))

    th = [threading.Thread(target=worker, args=(i,)) for i in range(2)]
    [t.start() for t in th]
    [t.join() for t in th]
    assert not errs and got == {0: 5, 1: 5}, errs


def test_verify_element_inside_a_stream_capture(H, golden):
    """The fork onto the ctx's side stream and the join are event dependencies, so the export can be captured into a HIP graph like any
    stream-ordered call: the replayed graph writes the image a plain call writes."""
    import ctypes
    from halo2_rsa_amd import _lib
    rsa = H.RSAChip(2048, 5)
    rng = random.Random(0x68327273 + 45)
    B = 16
    N = [rand_modulus(rng, 2048) for _ in range(B)]
    pk = rsa.assign_public_key(H.RSAPublicKey(H.UnassignedInteger.from_ints(N, 32, 64), H.Fix(65537)))
    sg = rsa.assign_signature(H.RSASignature(H.UnassignedInteger.from_ints([rng.randrange(n) for n in N], 32, 64)))
    res = rsa.verify_pkcs1v15_signature(pk, [rng.getrandbits(256) for _ in range(B)], sg)
    want = res.emit_advice(direct=True)          # (also: first use of every kernel and of the side stream, outside the capture)
    out = torch.zeros_like(want)
    sig, n, hashed = res.inputs
    chip = res.chip
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        _lib.check(_lib.lib().h2r_verify_emit_advice(chip._ctx, ctypes.byref(res.layout), sig.data_ptr(), n.data_ptr(), hashed.data_ptr(), res.powed.data_ptr(),
                                                     chip._flags(n, B) | _lib.H2R_ADVICE_DIRECT, res.trace.data_ptr(), res.workspace.data_ptr(), B,
                                                     res.status.data_ptr(), out.data_ptr(), out.shape[1], chip._stream()), "h2r_verify_emit_advice")
    assert int(out.max().item()) == 0            # captured, not run
    g.replay()
    torch.cuda.synchronize()
    assert torch.equal(out, want)
    out.zero_()
    g.replay()
    torch.cuda.synchronize()
    assert torch.equal(out, want)


def test_capture_on_one_stream_and_a_plain_emit_on_another(H, golden):
    """A capture in progress on stream A and a plain emit on stream B of the SAME ctx before EndCapture: the capturing call must not
    have pulled the ctx's side stream into its capture (it runs its row programs in order on A), so the plain call forks and joins as
    usual, the capture stays valid, and both images are right."""
    import ctypes
    from halo2_rsa_amd import _lib
    rsa = H.RSAChip(2048, 5)
    rng = random.Random(0x68327273 + 46)
    B = 8
    N = [rand_modulus(rng, 2048) for _ in range(B)]
    pk = rsa.assign_public_key(H.RSAPublicKey(H.UnassignedInteger.from_ints(N, 32, 64), H.Fix(65537)))
    sg = rsa.assign_signature(H.RSASignature(H.UnassignedInteger.from_ints([rng.randrange(n) for n in N], 32, 64)))
    res = rsa.verify_pkcs1v15_signature(pk, [rng.getrandbits(256) for _ in range(B)], sg)
    want = res.emit_advice(direct=True)
    out_graph, out_plain = torch.zeros_like(want), torch.zeros_like(want)
    sig, n, hashed = res.inputs
    chip = res.chip

    def emit(dst, stream_ptr):
        _lib.check(_lib.lib().h2r_verify_emit_advice(chip._ctx, ctypes.byref(res.layout), sig.data_ptr(), n.data_ptr(), hashed.data_ptr(), res.powed.data_ptr(),
                                                     chip._flags(n, B) | _lib.H2R_ADVICE_DIRECT, res.trace.data_ptr(), res.workspace.data_ptr(), B,
                                                     res.status.data_ptr(), dst.data_ptr(), dst.shape[1], stream_ptr), "h2r_verify_emit_advice")

    torch.cuda.synchronize()
    other = torch.cuda.Stream()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, capture_error_mode="relaxed"):
        emit(out_graph, chip._stream())
        emit(out_plain, ctypes.c_void_p(other.cuda_stream))       # a plain stream, while the capture is still open
        other.synchronize()                                        # ... and it RUNS while the capture is open
    assert torch.equal(out_plain, want)
    assert int(out_graph.max().item()) == 0
    g.replay()
    torch.cuda.synchronize()
    assert torch.equal(out_graph, want)


def test_config2_full_size_direct_image(H):
    """BASELINE config 2 (1,024 RSA-2048 signatures, e = 65537): the 12.4 GB image written directly equals the image of the
    records, every byte, compared on the device."""
    chip = H.BigIntChip(64, 2048)
    rng = random.Random(0x68327273 + 2)
    B = 1024
    N = [rand_modulus(rng, 2048) for _ in range(B)]
    X = [rng.randrange(n) for n in N]
    pres = chip.pow_mod_fixed_exp(chip.assign_integer(X), 65537, chip.assign_integer(N))
    a = pres.emit_advice()
    b = pres.emit_advice(direct=True)
    torch.cuda.synchronize()
    assert int(pres.status.max().item()) == 0
    step = 64
    for lo in range(0, B, step):
        assert torch.equal(a[lo:lo + step], b[lo:lo + step]), "elements %d.." % lo


@pytest.mark.parametrize("w,L,field,e_limbs,nb", [(64, 32, "bn254_fr", [0b10001, 0b00000, 0b00000, 0b00010], 5), (64, 16, "bn254_fq", [0x8000000000000003, 5], 64),
                                                   (32, 8, "pasta_fq", [0b1011011, 0b0000001, 0b1111111], 7), (32, 16, "pasta_fp", [0xC0000001], 32)])
def test_pow_var_advice_image(H, w, L, field, e_limbs, nb):
    """BigIntChip::pow_mod (RSAPubE::Var, big_integer/chip.rs:664-696; src/chip.rs:108-110) as advice rows: main_gate.to_bits of every
    exponent limb, acc = 1, per bit mul_mod / select / square_mod -- the GPU image (from the records and with H2R_ADVICE_DIRECT) equals
    the Python restatement run on the ORACLE's Var stream, the kinds are h2r_pow_row_kinds', the new kinds' fixed rows are the
    restatement's, and every to_bits / select row of the GPU image satisfies the main gate."""
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
    import pyref as R
    import advice_ref as AR
    from oracle_lib import Oracle
    from halo2_rsa_amd import _lib
    from halo2_rsa_amd._lib import lib
    chip = H.BigIntChip(w, w * L, field=field)
    P = R.FIELD_MODULI[field]
    o = Oracle(w, L)
    rng = random.Random(13 * w + L + nb)
    batch = 3
    N = [rand_modulus(rng, w * L, odd=(i != 1)) for i in range(batch)]
    X = [rng.randrange(n) for n in N]
    E = [list(e_limbs), [rng.getrandbits(nb) for _ in e_limbs], [0] * len(e_limbs)]
    e_dev = chip.assign_integer(H.UnassignedInteger(np.array(E, dtype=np.uint64 if w == 64 else np.uint32)))
    res = chip.pow_mod(chip.assign_integer(X), e_dev, chip.assign_integer(N), nb)
    torch.cuda.synchronize()
    assert res.status.cpu().tolist() == [0, 0, 0]
    pl = res.trace.pow_layout
    assert (pl.exp_limb_bits, pl.e_num_limbs) == (nb, len(e_limbs))
    total = int(lib().h2r_pow_advice_rows(chip._ctx, ctypes.byref(pl)))
    kinds = np.zeros(total, dtype=np.uint8)
    assert lib().h2r_pow_row_kinds(chip._ctx, ctypes.byref(pl), kinds.ctypes.data) == 0
    img = res.emit_advice().cpu().numpy().reshape(batch, total, 160)
    img_d = res.emit_advice(direct=True).cpu().numpy().reshape(batch, total, 160)
    assert np.array_equal(img, img_d)
    cfg = AR.LookupConfig(AR.range_lens(w, L, rsa=(w == 64)))
    la = H.LookupArgument(chip, rsa_chip=(w == 64))
    for i in range(batch):
        rc, out, st = o.pow_mod(o.limbs(X[i]), E[i], nb, o.limbs(N[i]))
        assert rc == 0
        im = AR.pow_var_image(o.p, [int(v) for v in o.limbs(X[i])], E[i], nb, [int(v) for v in o.limbs(N[i])], st, P, o.mul_mod_stream_bytes)
        assert im.kinds == kinds.tolist()
        want = AR.image_bytes(im)
        if not np.array_equal(img[i], want):
            bad = np.argwhere(img[i] != want)[0]
            pytest.fail("elem %d: row %d (kind %d) cell %d differs" % (i, int(bad[0]), int(kinds[int(bad[0])]), int(bad[1]) // 32))
    # the fixed side of the new kinds, and the gate on the GPU's own to_bits / select rows
    fixed = {}
    for k in sorted(set(kinds.tolist())):
        fr = _lib.H2RFixedRow()
        assert lib().h2r_advice_fixed_row(chip._ctx, ctypes.byref(la.cfg), k, ctypes.byref(fr)) == 0, k
        fixed[k] = fr.as_dict()
        if AR.ROW_BITS_COMPOSE <= k < AR.ROW_BITS_COMPOSE_LAST + 64:
            ref = AR.fixed_row_bits_compose(k)
            assert {nm: v % P for nm, v in ref.items() if nm in AR.FIXED_NAMES} == {nm: fixed[k][nm] for nm in AR.FIXED_NAMES}, k
            assert fixed[k]["tag_composition"] == 0 and fixed[k]["tag_overflow"] == 0
    per_limb = nb + (nb + 3) // 4 + 1
    head = len(e_limbs) * per_limb + 2
    cells = lambda r: [int.from_bytes(img[0, r, 32 * c:32 * c + 32].tobytes(), "little") for c in range(5)]
    check_rows = list(range(head)) + [r for r in range(head, total) if kinds[r] == AR.ROW_SELECT]
    for r in check_rows:
        nxt = cells(r + 1)[4] if r + 1 < total else 0
        assert AR.gate_residual(cells(r), nxt, fixed[int(kinds[r])], P) == 0, (r, int(kinds[r]))


def test_verify_element_with_a_variable_exponent_as_advice_rows(H, golden):
    """RSAChip::verify_pkcs1v15_signature with RSAPubE::Var (src/chip.rs:108-110): the whole element's image exists (it used to be
    H2R_E_UNSUPPORTED) -- [is_eq] [assert_in_field] [pow_mod: to_bits, acc = 1, mul_mod / select / square_mod per bit] [EM check] --
    its pow section is h2r_pow_trace_emit_advice's, and for the reference's KATs with e = 65537 as four 5-bit limbs... the result rows
    agree with the fixed-exponent element where both hold the same values (in-field and EM sections)."""
    from halo2_rsa_amd._lib import lib
    rsa = H.RSAChip(2048, 5)
    chip = rsa.bigint_chip()
    kats = golden["rsa_kats"]
    ns = [int(k["n"]) for k in kats]
    sigs = [int(k["sig"]) for k in kats]
    hashed = [int(k["hashed"]) for k in kats]
    e = 65537
    e_limbs = [[(e >> (5 * i)) & 31 for i in range(4)] for _ in kats]
    assert sum(v << (5 * i) for i, v in enumerate(e_limbs[0])) == e
    pk_var = rsa.assign_public_key(H.RSAPublicKey(H.UnassignedInteger.from_ints(ns, 32, 64), H.Var(H.UnassignedInteger(np.array(e_limbs, dtype=np.uint64)))))
    pk_fix = rsa.assign_public_key(H.RSAPublicKey(H.UnassignedInteger.from_ints(ns, 32, 64), H.Fix(e)))
    sg = rsa.assign_signature(H.RSASignature(H.UnassignedInteger.from_ints(sigs, 32, 64)))
    rv = rsa.verify_pkcs1v15_signature(pk_var, hashed, sg)
    rf = rsa.verify_pkcs1v15_signature(pk_fix, hashed, sg)
    total, sec = rv.advice_sections()
    _, sec_f = rf.advice_sections()
    rows = 3974
    assert sec[0] == 1 and sec[1] == 1532 and sec[3] == sec_f[3]
    assert sec[2] == 4 * (5 + 2 + 1) + 2 + 20 * (2 * rows + 32)
    assert rv.is_valid.cpu().tolist() == [1, 1, 0]
    img = rv.emit_advice().cpu().numpy().reshape(3, total, 160)
    imf = rf.emit_advice().cpu().numpy().reshape(3, sum(sec_f), 160)
    assert np.array_equal(img[:, :1 + sec[1]], imf[:, :1 + sec_f[1]])            # is_eq seed + assert_in_field
    assert np.array_equal(img[:, total - sec[3]:], imf[:, sum(sec_f) - sec_f[3]:])   # the encoded-message check (same powed)
    assert torch.equal(rv.emit_advice(direct=True), rv.emit_advice())
    kinds = rv.row_kinds()
    pk = np.zeros(sec[2], dtype=np.uint8)
    assert lib().h2r_pow_row_kinds(chip._ctx, ctypes.byref(rv.layout.pow), pk.ctypes.data) == 0
    assert np.array_equal(kinds[1 + sec[1]:1 + sec[1] + sec[2]], pk)
