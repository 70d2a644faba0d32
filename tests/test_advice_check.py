"""h2r_advice_check: the device-side MockProver.  Every reference test is `MockProver::run(k, &circuit, ..).verify()` (src/chip.rs:338-345,
667; big_integer/chip.rs:1454-1458; examples/rsa_example.rs:207-212); this is its counterpart for an image in HBM: gate, lookup and copy
checks of every row of every element, independent of the kernels that wrote the image.  Green on valid images (every representation,
record-read and direct, fixed and variable exponent, whole verify elements, BASELINE config 2 at full size) and red on EVERY single-cell
corruption (the instrument must not be vacuous)."""
import ctypes
import json
import os
import random
import sys

import numpy as np
import pytest

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))


@pytest.fixture(scope="module")
def H():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    import halo2_rsa_amd as H_
    return H_


def rand_modulus(rng, bits, odd=True):
    n = rng.getrandbits(bits) | (1 << (bits - 1))
    return n | 1 if odd else n & ~1


def _kinds(chip, pl):
    from halo2_rsa_amd._lib import lib
    n = int(lib().h2r_pow_advice_rows(chip._ctx, ctypes.byref(pl)))
    k = np.zeros(n, dtype=np.uint8)
    assert lib().h2r_pow_row_kinds(chip._ctx, ctypes.byref(pl), k.ctypes.data) == 0
    return k


REPRS = [dict(), dict(columns=True), dict(montgomery=True), dict(columns=True, montgomery=True)]


@pytest.mark.parametrize("w,L,field", [(64, 32, "bn254_fr"), (64, 16, "pasta_fp"), (32, 16, "pasta_fq"), (32, 128, "bn254_fq")])
@pytest.mark.parametrize("repr_kw", REPRS)
def test_valid_pow_images_pass_with_copies(H, w, L, field, repr_kw):
    """A fixed-exponent pow element, record-read and direct: no violated gate, lookup or copy pair; even / small / all-ones operands among them."""
    chip = H.BigIntChip(w, w * L, field=field, **repr_kw)
    look = H.LookupArgument(chip, rsa_chip=False)
    rng = random.Random(w + L)
    batch, e = 4, 0b1101
    N = [rand_modulus(rng, w * L, odd=(i != 1)) for i in range(batch)]
    X = [rng.randrange(n) for n in N]
    X[2] = N[2] - 1
    X[3] = 0
    x, n = chip.assign_integer(X), chip.assign_integer(N)
    res = chip.pow_mod_fixed_exp(x, e, n)
    pl = res.trace.pow_layout
    kinds = _kinds(chip, pl)
    copies = chip.pow_copy_map(pl, e)
    for direct in (False, True):
        img = res.emit_advice(direct=direct)
        bad, first = chip.advice_check(kinds, img, batch, status=res.status, copies=copies, src_a=x, src_n=n, lookup=look)
        assert bad.cpu().tolist() == [0] * batch, (direct, bad.cpu().tolist(), [hex(v) for v in first.cpu().tolist()])


def _cell_view(img_host, repr_kw, rows, elem, row, col):
    """the 32 bytes of one cell of a packed image (numpy view)"""
    if repr_kw.get("columns"):
        o = col * rows * 32 + row * 32
    else:
        o = row * 160 + col * 32
    return img_host[elem, o:o + 32]


@pytest.mark.parametrize("repr_kw", [dict(), dict(columns=True, montgomery=True)])
def test_every_single_cell_corruption_is_flagged(H, repr_kw):
    """One cell changed anywhere -- mul rows, range rows (a sub-limb pushed out of its table, a remainder), eq_b, the carry chain, flags,
    constants, a cell that only a COPY pair protects -- and the element is flagged with the right code; its neighbours stay clean."""
    from halo2_rsa_amd._lib import lib
    chip = H.BigIntChip(64, 2048, **repr_kw)
    look = H.LookupArgument(chip, rsa_chip=False)
    rng = random.Random(4)
    batch, e = 3, 3
    N = [rand_modulus(rng, 2048) for _ in range(batch)]
    X = [rng.randrange(n) for n in N]
    x, n = chip.assign_integer(X), chip.assign_integer(N)
    res = chip.pow_mod_fixed_exp(x, e, n)
    pl = res.trace.pow_layout
    kinds = _kinds(chip, pl)
    copies = chip.pow_copy_map(pl, e)
    good = res.emit_advice(direct=True)
    bad, _ = chip.advice_check(kinds, good, batch, copies=copies, src_a=x, src_n=n, lookup=look)
    assert bad.cpu().tolist() == [0, 0, 0]
    rows = len(kinds)
    rec_rows = int(lib().h2r_advice_rows(chip._ctx))
    host = good.cpu().numpy()
    P = __import__("pyref").FIELD_MODULI["bn254_fr"]
    R = 1 << 256
    # targets: (row in the element, column); rows of record 1 (the second mul_mod) unless stated
    b1 = 2 + rec_rows
    L = 32
    r_T3, mul_rows = 4 * L, 63 + L * L
    r_T5 = r_T3 + 2 * mul_rows
    r_T6 = r_T5 + L + 4
    targets = [(0, 0), (1, 0),                                     # the constants of acc = 1
               (b1 + 0, 0), (b1 + 0, 4), (b1 + 1, 2), (b1 + 1, 4), (b1 + 127, 0),   # range rows of q / r limbs: sub-limbs, remainders
               (b1 + r_T3 + 1, 0), (b1 + r_T3 + 1, 1), (b1 + r_T3 + 1, 3), (b1 + r_T3 + 559, 2), (b1 + r_T3 + mul_rows + 4, 3),   # mul rows (column 31's last, column 1's second)
               (b1 + r_T5 + 3, 0), (b1 + r_T5 + 3, 1), (b1 + r_T5 + 3, 2),          # eq_b
               (b1 + r_T5 + L, 0), (b1 + r_T5 + L + 3, 1),                          # is_equal_muled preamble: 2^w, the bit
               (b1 + r_T6 + 0, 2), (b1 + r_T6 + 1, 2), (b1 + r_T6 + 2, 0), (b1 + r_T6 + 3, 0), (b1 + r_T6 + 8, 0), (b1 + r_T6 + 9, 0),
               (b1 + r_T6 + 14, 0), (b1 + r_T6 + 17, 2), (b1 + r_T6 + 18, 0), (b1 + r_T6 + 19, 1), (b1 + r_T6 + 20, 4),   # carry range rows
               (b1 + r_T6 + 26 * 5 + 22, 2), (rows - 2, 2), (rows - 2, 0),    # the last column's closing `and`
               (rows - 1, 0)]                                                  # assert_equal_muled's assert_one(eq_bit), the record's last row (chip.rs:1062)
    assert len(targets) >= 20
    codes = set()
    for (row, col) in targets:
        h2 = host.copy()
        cell = _cell_view(h2, repr_kw, rows, 1, row, col)
        v = int.from_bytes(cell.tobytes(), "little")
        nv = (v + (R % P if repr_kw.get("montgomery") else 1)) % P      # the cell's integer + 1
        cell[:] = np.frombuffer(nv.to_bytes(32, "little"), dtype=np.uint8)
        dev = torch.from_numpy(h2).cuda()
        bad, first = chip.advice_check(kinds, dev, batch, copies=copies, src_a=x, src_n=n, lookup=look)
        b = bad.cpu().tolist()
        assert b[0] == 0 and b[2] == 0 and b[1] > 0, ((row, col), b)
        codes.add(int(first.cpu()[1]) & 0xff)
    assert {1, 2, 3} <= codes, codes          # gate, lookup and copy violations all occurred
    # a cell >= p
    h2 = host.copy()
    _cell_view(h2, repr_kw, rows, 1, b1 + r_T3 + 9, 3)[:] = 0xFF
    bad, first = chip.advice_check(kinds, torch.from_numpy(h2).cuda(), batch, copies=copies, src_a=x, src_n=n, lookup=look)
    assert bad.cpu().tolist()[1] > 0 and (int(first.cpu()[1]) & 0xff) == 5
    # a copy pair against an operand: the base x changed behind the image's back
    x2 = chip.assign_integer([X[0], X[1] ^ 1, X[2]])
    bad, first = chip.advice_check(kinds, good, batch, copies=copies, src_a=x2, src_n=n, lookup=look)
    assert bad.cpu().tolist()[0] == 0 and bad.cpu().tolist()[1] > 0 and (int(first.cpu()[1]) & 0xff) == 3


@pytest.mark.parametrize("repr_kw", [dict(), dict(columns=True, montgomery=True)])
def test_verify_element_var_arm_and_layout(H, repr_kw):
    """A whole verify_pkcs1v15_signature element (seed, assert_in_field, pow, encoded-message check; RangeChip's 4-bit table included),
    the variable-exponent arm (to_bits, select rows), and an image permuted to a custom layout checked UNDER that layout."""
    import hashlib
    from halo2_rsa_amd import _lib
    from halo2_rsa_amd._lib import lib
    with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "halo2_rsa_golden.json")) as f:
        kats = json.load(f)["rsa_kats"]
    ns, sigs = [int(k["n"]) for k in kats], [int(k["sig"]) for k in kats]
    h = int.from_bytes(hashlib.sha256(b"hello world").digest(), "big")
    hashed = [[(h >> (64 * j)) & 0xFFFFFFFFFFFFFFFF for j in range(4)]] * 3
    rsa = H.RSAChip(2048, 5, **repr_kw)
    chip = rsa.bigint_chip()
    look = H.LookupArgument(chip, rsa_chip=True)
    pk = rsa.assign_public_key(H.RSAPublicKey(H.UnassignedInteger.from_ints(ns, 32, 64), H.Fix(65537)))
    sig = rsa.assign_signature(H.RSASignature(H.UnassignedInteger.from_ints(sigs, 32, 64)))
    hd = torch.tensor(np.array(hashed, dtype=np.uint64).view(np.int64), device="cuda")
    res = rsa.verify_pkcs1v15_signature(pk, hd, sig)
    kinds = res.row_kinds()
    total, sec = res.advice_sections()
    copies = chip.pow_copy_map(res.layout.pow, 65537, row_offset=sec[0] + sec[1])
    for direct in (False, True):
        img = res.emit_advice(direct=direct)
        bad, first = chip.advice_check(kinds, img, 3, copies=copies, src_a=sig.c if hasattr(sig, "c") else sig, src_n=pk.n, lookup=look)
        # KAT1 / KAT2 are valid signatures; BAD's encoded message differs, which its is_valid bit records -- the ASSIGNMENT still satisfies every gate
        assert bad.cpu().tolist() == [0, 0, 0], (direct, bad.cpu().tolist(), [hex(v) for v in first.cpu().tolist()])
    # a custom layout, applied and checked
    lay = _lib.H2RAdviceLayout()
    ks = (ctypes.c_uint8 * 2)(6, 7)
    cols = ((ctypes.c_uint8 * 5) * 2)((1, 0, 2, 3, 4), (1, 0, 2, 3, 4))
    assert lib().h2r_advice_layout_custom(chip._ctx, ks, cols, 2, ctypes.byref(lay)) == 0
    kd = torch.tensor(kinds, device="cuda")
    perm = img.clone()
    assert lib().h2r_advice_apply_layout(chip._ctx, ctypes.byref(lay), kd.data_ptr(), len(kinds), perm.data_ptr(), perm.shape[1], 3, None,
                                         chip._stream()) == 0
    bad, _ = chip.advice_check(kinds, perm, 3, copies=copies, src_a=sig.c if hasattr(sig, "c") else sig, src_n=pk.n, lookup=look, layout=lay)
    assert bad.cpu().tolist() == [0, 0, 0]
    # a hand-filled layout that is not a permutation is refused (it would index past a row)
    lay.column_of[6][0] = 7
    assert lib().h2r_advice_apply_layout(chip._ctx, ctypes.byref(lay), kd.data_ptr(), len(kinds), perm.data_ptr(), perm.shape[1], 3, None,
                                         chip._stream()) == _lib.H2R_E_SHAPE
    # a hand-filled PERMUTATION that moves a off column 0 on a decomposition row (kind 32 = H2R_ROW_RANGE_LIMB): h2r_advice_layout_custom refuses
    # it, and so do the two exports whose lookup passes read physical columns 0..3 / 0 of such rows (they would report false violations)
    lay2 = _lib.H2RAdviceLayout()
    assert lib().h2r_advice_layout_default(ctypes.byref(lay2)) == 0
    lay2.column_of[32][0], lay2.column_of[32][1] = 1, 0
    k32 = (ctypes.c_uint8 * 1)(32)
    c32 = ((ctypes.c_uint8 * 5) * 1)((1, 0, 2, 3, 4))
    assert lib().h2r_advice_layout_custom(chip._ctx, k32, c32, 1, ctypes.byref(_lib.H2RAdviceLayout())) == _lib.H2R_E_SHAPE
    with pytest.raises(Exception):
        chip.advice_check(kinds, img, 3, lookup=look, layout=lay2)
    hist = torch.zeros(3 * 5 * look.cfg.n_rows, dtype=torch.int32, device="cuda")
    assert lib().h2r_lookup_hist_advice(chip._ctx, ctypes.byref(look.cfg), ctypes.byref(lay2), kd.data_ptr(), len(kinds), img.data_ptr(), img.shape[1], 3,
                                        None, hist.data_ptr(), chip._stream()) == _lib.H2R_E_SHAPE
    # the variable-exponent arm
    x = chip.assign_integer(sigs)
    pkv = rsa.assign_public_key(H.RSAPublicKey(H.UnassignedInteger.from_ints(ns, 32, 64), H.Var(H.UnassignedInteger.from_ints([19, 31, 1], 1, 64))))
    rv = rsa.modpow_public_key(x, pkv)
    plv = rv.trace.pow_layout
    kv = _kinds(chip, plv)
    for direct in (False, True):
        bad, first = chip.advice_check(kv, rv.emit_advice(direct=direct), 3, lookup=look)
        assert bad.cpu().tolist() == [0, 0, 0], (direct, bad.cpu().tolist(), [hex(v) for v in first.cpu().tolist()])


def test_config2_full_size_direct_image(H):
    """BASELINE config 2 at full size (1,024 RSA-2048 signatures, e = 65537): the 12.4 GB direct image, all 77 M rows and 0.25 G copy pairs."""
    chip = H.BigIntChip(64, 2048)
    look = H.LookupArgument(chip, rsa_chip=False)
    rng = random.Random(2)
    batch, e = 1024, 65537
    N = [rand_modulus(rng, 2048) for _ in range(batch)]
    X = [rng.randrange(n) for n in N]
    x, n = chip.assign_integer(X), chip.assign_integer(N)
    T = chip.pow_fixed_layout(e).num_mul_mods
    res = chip.pow_mod_fixed_exp(x, e, n, want_trace=False, workspace=torch.empty(chip.workspace_bytes(batch, T), dtype=torch.uint8, device="cuda"))
    img = res.emit_advice(direct=True)
    pl = res.pow_layout
    kinds = _kinds(chip, pl)
    copies = chip.pow_copy_map(pl, e)
    bad, first = chip.advice_check(kinds, img, batch, status=res.status, copies=copies, src_a=x, src_n=n, lookup=look)
    assert int(bad.sum().item()) == 0, (int((bad != 0).sum().item()), hex(int(first[bad != 0][0].item())))
    assert int(res.status.max().item()) == 0
