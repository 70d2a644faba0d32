"""h2r_lookup_hist_advice: the lookup multiplicities of a witness that has no records, counted from its advice image.  The reference never
counts anything -- halo2's prover reads the lookup-enabled cells of the assigned columns (benches/bench.rs:141-142, 321-329: load_table +
create_proof); the record-based exports (h2r_lookup_hist_verify / _records / _fresh_op, pinned against the Python restatement of
permute_expression_pair on the oracle's cells in tests/test_lookup_gpu.py) count the same cells from the trace.  Here: the image's count must
be the trace's count, for whole verify elements (RangeChip's 4-bit table included) and modpow elements, in every representation, under a
custom layout, and for the records-free pipelined image; and A' / S' built from it are byte-identical."""
import ctypes
import os
import random
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
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


@pytest.mark.gpu
@pytest.mark.parametrize("repr_kw", [dict(), dict(columns=True), dict(montgomery=True), dict(columns=True, montgomery=True)])
def test_image_multiplicities_are_the_trace_multiplicities(H, golden, repr_kw):
    from halo2_rsa_amd import _lib
    from halo2_rsa_amd._lib import lib
    rsa = H.RSAChip(2048, 5, **repr_kw)
    chip = rsa.bigint_chip()
    la = H.LookupArgument(chip, rsa_chip=True)
    kats = golden["rsa_kats"]
    rng = random.Random(5)
    B = 6
    ns = [int(k["n"]) for k in kats] + [rand_modulus(rng, 2048) for _ in range(B - 3)]
    sigs = [int(k["sig"]) for k in kats] + [rng.randrange(n) for n in ns[3:]]
    hashed = [int(k["hashed"]) for k in kats] + [rng.getrandbits(256) for _ in range(B - 3)]
    sigs[4] = ns[4] + 1                                           # not in the field: a status, no pow / EM rows
    pk = rsa.assign_public_key(H.RSAPublicKey(H.UnassignedInteger.from_ints(ns, 32, 64), H.Fix(65537)))
    sg = rsa.assign_signature(H.RSASignature(H.UnassignedInteger.from_ints(sigs, 32, 64)))
    res = rsa.verify_pkcs1v15_signature(pk, _hashed_tensor(hashed), sg)
    ok = (res.status == 0)
    want = la.hist_verify(res, la.new_hist(B))
    kinds = res.row_kinds()
    for direct in (False, True):
        got = la.hist_advice(kinds, res.emit_advice(direct=direct), B, la.new_hist(B), status=res.status)
        torch.cuda.synchronize()
        assert torch.equal(got[ok], want[ok]), direct
        assert int(got[4].sum()) == 0                             # the skipped element (hist_verify counts its in-field witness: a trace exists)
    assert int(want[0].sum()) > 20000 and int(want[0, 4].sum()) > 1000        # composition and overflow lookups both present
    # ADDED to, like every hist export
    twice = la.hist_advice(kinds, res.emit_advice(), B, got.clone(), status=res.status)
    assert torch.equal(twice[ok], 2 * want[ok])
    # under a custom layout: the permuted image with its layout counts the same
    if not repr_kw:
        lay = _lib.H2RAdviceLayout()
        ks = (ctypes.c_uint8 * 2)(6, 7)
        cols = ((ctypes.c_uint8 * 5) * 2)((1, 0, 2, 3, 4), (1, 0, 2, 3, 4))
        assert lib().h2r_advice_layout_custom(chip._ctx, ks, cols, 2, ctypes.byref(lay)) == 0
        perm = res.emit_advice()
        kd = torch.tensor(kinds, device="cuda")
        assert lib().h2r_advice_apply_layout(chip._ctx, ctypes.byref(lay), kd.data_ptr(), len(kinds), perm.data_ptr(), perm.shape[1], B, None, chip._stream()) == 0
        got_l = la.hist_advice(kd, perm, B, la.new_hist(B), status=res.status, layout=lay)
        assert torch.equal(got_l[ok], want[ok])
    # the records-free pipelined image, and A' / S' from its multiplicities
    pipe = H.Pipeline(chip, 2, 2)
    vl = pipe.verify_compact_layout(65537)
    rows = len(kinds)
    img = torch.empty((B, chip.image_bytes(rows)), dtype=torch.uint8, device="cuda")
    wit = torch.zeros((B, vl.elem_stride), dtype=torch.uint8, device="cuda")
    ws = torch.empty(chip.workspace_bytes(B, vl.pow.num_mul_mods), dtype=torch.uint8, device="cuda")
    powed = torch.zeros((B, chip.num_limbs), dtype=torch.int64, device="cuda")
    valid, st = torch.zeros(B, dtype=torch.uint8, device="cuda"), torch.zeros(B, dtype=torch.uint8, device="cuda")
    pipe.verify_pkcs1v15_advice(chip.assign_integer(sigs), 65537, chip.assign_integer(ns), _hashed_tensor(hashed), wit, ws, powed, valid, st, img)
    pipe.join()
    pipe.close()
    got_p = la.hist_advice(kinds, img, B, la.new_hist(B), status=st)
    torch.cuda.synchronize()
    assert torch.equal(got_p[ok], want[ok])
    sel = torch.nonzero(ok).flatten()[:3]
    P = {"bn254_fr": 21888242871839275222246405745257275088548364400416034343698204186575808495617}["bn254_fr"]
    thetas = [rng.randrange(P) for _ in range(len(sel))]
    usable = (1 << 17) - 6
    a1, s1, e1 = la.permuted_columns(want[sel].contiguous(), thetas, usable)
    a2, s2, e2 = la.permuted_columns(got_p[sel].contiguous(), thetas, usable)
    torch.cuda.synchronize()
    assert not e1.cpu().numpy().any() and torch.equal(a1, a2) and torch.equal(s1, s2)


@pytest.mark.gpu
def test_modpow_image_multiplicities(H):
    """A modpow_public_key element ([assert_in_field rows][pow rows], BigIntChip's table: no 4-bit length): records + in-field witness counted
    from the traces = the image's count; 32-bit limbs too (sub-limbs of 4 bits, carries split into 5-bit sub-limbs)."""
    from halo2_rsa_amd import _lib
    from halo2_rsa_amd._lib import lib
    for w, bits in ((64, 2048), (32, 1024)):
        chip = H.BigIntChip(w, bits)
        la = H.LookupArgument(chip, rsa_chip=False)
        rng = random.Random(w)
        B = 4
        N = [rand_modulus(rng, bits) for _ in range(B)]
        X = [rng.randrange(n) for n in N]
        res = chip.pow_mod_fixed_exp(chip.assign_integer(X), 65537, chip.assign_integer(N), check_in_field=True)
        want = la.hist_records(res.trace, la.new_hist(B), status=res.status)
        la.hist_fresh_op("is_in_field", res.in_field.buf, res.in_field.elem_stride, B, want)
        pl = res.trace.pow_layout
        k_if = chip.fresh_op_row_kinds(_lib.FRESH_OPS.index("is_in_field"), assert_one=True)
        k_pow = np.zeros(int(lib().h2r_pow_advice_rows(chip._ctx, ctypes.byref(pl))), dtype=np.uint8)
        assert lib().h2r_pow_row_kinds(chip._ctx, ctypes.byref(pl), k_pow.ctypes.data) == 0
        got = la.hist_advice(np.concatenate([k_if, k_pow]), res.emit_modpow_advice(), B, la.new_hist(B), status=res.status)
        torch.cuda.synchronize()
        assert torch.equal(got, want), (w, bits)
        assert int(got.sum()) > 0
