"""What a layouter backend needs besides the cells (VERDICT r3 Missing #4 / #5): the COPY MAP of the advice image -- which input
cell of which row is an equality-constrained copy of which earlier cell (maingate ties `AssignedValue::from(a.limb(j))` into
main_gate.mul_add, big_integer/chip.rs:406-408; `carry` across the steps of is_equal_muled, :861; ...) -- and the LAYOUT AS DATA:
the physical column of every cell of every row kind, overridable without recompiling a kernel, since the placement is third-party
code restated from recollection."""
import ctypes
import os
import random
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def _host_ctx(w, bits, field=0):
    from halo2_rsa_amd import _lib
    L = _lib.lib()
    pr = _lib.H2RParams()
    pr.limb_width, pr.bits_len, pr.field, pr.device = w, bits, field, -1
    ctx = ctypes.c_void_p()
    assert L.h2r_ctx_create(ctypes.byref(pr), ctypes.byref(ctx)) == 0
    return L, ctx


@pytest.mark.parametrize("w,L_", [(64, 32), (32, 128), (64, 4), (32, 8)])
def test_copy_map_is_well_formed(w, L_):
    """Every pair names a cell of the record (or limb j < num_limbs of an operand), sources come BEFORE their copies (a value is
    assigned before it is copied), no input cell has two origins, and the count is the closed form of the row table."""
    from halo2_rsa_amd import _lib
    L, ctx = _host_ctx(w, w * L_)
    rows = int(L.h2r_advice_rows(ctx))
    n = int(L.h2r_advice_copy_map(ctx, None, 0))
    buf = (_lib.H2RCopy * n)()
    assert int(L.h2r_advice_copy_map(ctx, buf, n)) == n
    C = 2 * L_ - 1
    assert n == 2 * 3 * L_ * L_ + 2 * L_ + C * 33 + 1   # three inputs per mul_add row, two per eq_b row, 33 per is_equal_muled column, the closing assert_one's one
    seen = set()
    kinds = np.zeros(rows, dtype=np.uint8)
    assert L.h2r_advice_row_kinds(ctx, kinds.ctypes.data) == 0
    for c in buf:
        assert c.row < rows and c.col < 5 and (c.row, c.col) not in seen
        seen.add((c.row, c.col))
        if c.src_row >= 0xFFFFFF00:
            assert c.src_row in (_lib.H2R_COPY_SRC_A, _lib.H2R_COPY_SRC_B, _lib.H2R_COPY_SRC_N) and c.src_col < L_
        else:
            assert c.src_row < c.row and c.src_col < 5
    L.h2r_ctx_destroy(ctx)


def test_pow_operand_sources_follow_the_reference_loop():
    """pow_mod_fixed_exp (big_integer/chip.rs:729-740): per exponent bit the squaring of `squared`, then -- bit set -- acc * (the value
    BEFORE that squaring).  e = 0b1011: sq0(x, x); mul(1, x); sq1(r0, r0); mul(r1, r0); sq2(r2, r2); sq3(r4, r4); mul(r3, r4)."""
    from halo2_rsa_amd import _lib
    L, ctx = _host_ctx(64, 2048)
    pl = _lib.H2RPowLayout()
    e = (0b1011).to_bytes(1, "little")
    assert L.h2r_pow_fixed_layout(ctx, e, 1, ctypes.byref(pl)) == 0 and pl.num_mul_mods == 7
    a, b = (ctypes.c_int32 * 7)(), (ctypes.c_int32 * 7)()
    assert L.h2r_pow_operand_sources(ctx, ctypes.byref(pl), e, 1, a, b) == 0
    X, ONE = _lib.H2R_SRC_X, _lib.H2R_SRC_ONE
    assert list(a) == [X, ONE, 0, 1, 2, 4, 3] and list(b) == [X, X, 0, 0, 2, 4, 4]
    L.h2r_ctx_destroy(ctx)


def test_layout_descriptor_validation_and_fixed_rows():
    """h2r_advice_layout_custom refuses what would change the meaning of a row: a non-permutation, a product a * b split over the
    gate's two column pairs, a decompose row whose running value leaves column e; h2r_advice_fixed_row_ex moves the selectors with
    the cells (and the a * b product to c * d when the pair moves)."""
    from halo2_rsa_amd import _lib
    L, ctx = _host_ctx(64, 2048)
    lay = _lib.H2RAdviceLayout()
    assert L.h2r_advice_layout_default(ctypes.byref(lay)) == 0 and lay.version == 1
    assert all(list(lay.column_of[k]) == [0, 1, 2, 3, 4] for k in range(256))

    def custom(kind, cols):
        k = (ctypes.c_uint8 * 1)(kind)
        c = ((ctypes.c_uint8 * 5) * 1)((ctypes.c_uint8 * 5)(*cols))
        out = _lib.H2RAdviceLayout()
        return L.h2r_advice_layout_custom(ctx, k, c, 1, ctypes.byref(out)), out
    MUL_ADD, SUB, RANGE0, SELECT = 6, 8, 32, 15
    assert custom(MUL_ADD, [0, 1, 2, 2, 4])[0] != 0                 # not a permutation
    assert custom(MUL_ADD, [0, 2, 1, 3, 4])[0] != 0                 # a * b split over the pairs
    assert custom(RANGE0, [0, 1, 2, 4, 3])[0] != 0                  # the running value must stay in e
    assert custom(SELECT, [1, 0, 3, 2, 4])[0] == 0                  # pairs swapped inside: fine
    assert custom(SELECT, [2, 3, 1, 0, 4])[0] == 0                  # the two products trade places
    rc, lay2 = custom(MUL_ADD, [2, 3, 0, 1, 4])
    assert rc == 0 and list(lay2.column_of[MUL_ADD]) == [2, 3, 0, 1, 4] and list(lay2.column_of[SUB]) == [0, 1, 2, 3, 4]
    f0, f1 = _lib.H2RFixedRow(), _lib.H2RFixedRow()
    assert L.h2r_advice_fixed_row(ctx, None, MUL_ADD, ctypes.byref(f0)) == 0
    assert L.h2r_advice_fixed_row_ex(ctx, None, ctypes.byref(lay2), MUL_ADD, ctypes.byref(f1)) == 0
    d0, d1 = f0.as_dict(), f1.as_dict()
    assert d0["s_mul_ab"] == 1 and d0["s_mul_cd"] == 0 and d1["s_mul_ab"] == 0 and d1["s_mul_cd"] == 1
    assert (d1["sa"], d1["sb"], d1["sc"], d1["sd"]) == (d0["sc"], d0["sd"], d0["sa"], d0["sb"])
    L.h2r_ctx_destroy(ctx)


@pytest.mark.gpu
@pytest.mark.parametrize("w,L_,field", [(64, 32, "bn254_fr"), (32, 16, "pasta_fq")])
def test_copies_hold_equal_values_in_the_gpu_image(w, L_, field):
    """Every pair of the copy map names two cells of the GPU's image with the SAME value (operand limbs for the external sources),
    and in a pow element the operand cells of record t equal the r-limb cells of the record h2r_pow_operand_sources names (x, or
    the constant 1 in front)."""
    torch = pytest.importorskip("torch")
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    import halo2_rsa_amd as H
    from halo2_rsa_amd import _lib
    from halo2_rsa_amd._lib import lib
    chip = H.BigIntChip(w, w * L_, field=field)
    rng = random.Random(w + L_)
    N = [rng.getrandbits(w * L_) | (1 << (w * L_ - 1)) | 1 for _ in range(2)]
    X = [rng.randrange(n) for n in N]
    e = 0b1011
    pres = chip.pow_mod_fixed_exp(chip.assign_integer(X), e, chip.assign_integer(N))
    rows = int(lib().h2r_advice_rows(chip._ctx))
    img = pres.emit_advice(direct=True).cpu().numpy().reshape(2, -1, 160)[:, 2:]      # (behind the two constant rows)
    pre = pres.emit_advice(direct=True).cpu().numpy().reshape(2, -1, 160)[:, :2]
    T = pres.trace.num_mul_mods
    n = int(lib().h2r_advice_copy_map(chip._ctx, None, 0))
    cm = (_lib.H2RCopy * n)()
    lib().h2r_advice_copy_map(chip._ctx, cm, n)
    eb = e.to_bytes(1, "little")
    a_src, b_src = (ctypes.c_int32 * T)(), (ctypes.c_int32 * T)()
    assert lib().h2r_pow_operand_sources(chip._ctx, ctypes.byref(pres.trace.pow_layout), eb, 1, a_src, b_src) == 0
    cell = lambda el, t, r, c: int.from_bytes(img[el, t * rows + r, 32 * c:32 * c + 32].tobytes(), "little")
    B = 1 << w
    limbs = lambda v: [(v >> (w * i)) & (B - 1) for i in range(L_)]

    def operand(el, src, j):   # limb j of the value a source code names
        if src == _lib.H2R_SRC_X:
            return limbs(X[el])[j]
        if src == _lib.H2R_SRC_ONE:
            return int.from_bytes(pre[el, 0 if j == 0 else 1, :32].tobytes(), "little")
        return cell(el, src, 2 * (L_ + j), 4)                      # r limb j of record `src`: column e of its range assign's first row
    for el in range(2):
        for t in range(T):
            for c in cm:
                got = cell(el, t, c.row, c.col)
                if c.src_row == _lib.H2R_COPY_SRC_A:
                    want = operand(el, a_src[t], c.src_col)
                elif c.src_row == _lib.H2R_COPY_SRC_B:
                    want = operand(el, b_src[t], c.src_col)
                elif c.src_row == _lib.H2R_COPY_SRC_N:
                    want = limbs(N[el])[c.src_col]
                else:
                    want = cell(el, t, c.src_row, c.src_col)
                assert got == want, (el, t, c.row, c.col, c.src_row, c.src_col)
    assert operand(0, _lib.H2R_SRC_ONE, 0) == 1 and operand(0, _lib.H2R_SRC_ONE, 3) == 0


@pytest.mark.gpu
def test_a_custom_layout_moves_cells_and_selectors_together():
    """Permute columns through the descriptor -- mul_add's (a, b) <-> (c, d), sub's a <-> b, assign_bit's cells to (c, d, a), is_zero's
    rows -- apply it to an emitted pow image on the device, and check that every row still satisfies the main-gate equation with
    h2r_advice_fixed_row_ex's selectors; rows of untouched kinds are byte-identical, and the permuted ones hold the same cells elsewhere."""
    torch = pytest.importorskip("torch")
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
    import pyref as R
    import advice_ref as AR
    import halo2_rsa_amd as H
    from halo2_rsa_amd import _lib
    from halo2_rsa_amd._lib import lib
    chip = H.BigIntChip(64, 1024)
    P = R.FIELD_MODULI["bn254_fr"]
    rng = random.Random(9)
    N = [rng.getrandbits(1024) | (1 << 1023) | 1 for _ in range(2)]
    X = [rng.randrange(n) for n in N]
    pres = chip.pow_mod_fixed_exp(chip.assign_integer(X), 5, chip.assign_integer(N))
    pl = pres.trace.pow_layout
    total = int(lib().h2r_pow_advice_rows(chip._ctx, ctypes.byref(pl)))
    kinds = np.zeros(total, dtype=np.uint8)
    assert lib().h2r_pow_row_kinds(chip._ctx, ctypes.byref(pl), kinds.ctypes.data) == 0
    before = pres.emit_advice(direct=True)
    img = before.clone()
    perms = {AR.ROW_MUL_ADD: [2, 3, 0, 1, 4], AR.ROW_SUB: [1, 0, 2, 3, 4], AR.ROW_BIT: [2, 3, 0, 1, 4], AR.ROW_ISZERO_INV: [3, 2, 1, 0, 4],
             AR.ROW_ADD: [0, 1, 4, 3, 2]}
    ks = (ctypes.c_uint8 * len(perms))(*perms.keys())
    cols = ((ctypes.c_uint8 * 5) * len(perms))(*[(ctypes.c_uint8 * 5)(*v) for v in perms.values()])
    lay = _lib.H2RAdviceLayout()
    assert lib().h2r_advice_layout_custom(chip._ctx, ks, cols, len(perms), ctypes.byref(lay)) == 0
    kd = torch.from_numpy(kinds).cuda()
    assert lib().h2r_advice_apply_layout(chip._ctx, ctypes.byref(lay), kd.data_ptr(), total, img.data_ptr(), img.shape[1], 2, pres.status.data_ptr(),
                                         chip._stream()) == 0
    torch.cuda.synchronize()
    a, b = before.cpu().numpy().reshape(2, total, 5, 32), img.cpu().numpy().reshape(2, total, 5, 32)
    for k in sorted(set(kinds.tolist())):
        sel = kinds == k
        pm = perms.get(k, [0, 1, 2, 3, 4])
        for c in range(5):
            assert np.array_equal(b[:, sel, pm[c]], a[:, sel, c]), (k, c)
    la = H.LookupArgument(chip, rsa_chip=True)
    fixed = {}
    for k in sorted(set(kinds.tolist())):
        fr = _lib.H2RFixedRow()
        assert lib().h2r_advice_fixed_row_ex(chip._ctx, ctypes.byref(la.cfg), ctypes.byref(lay), k, ctypes.byref(fr)) == 0
        fixed[k] = fr.as_dict()
    cells = lambda r: [int.from_bytes(b[0, r, c].tobytes(), "little") for c in range(5)]
    for r in range(total):
        nxt = cells(r + 1)[4] if r + 1 < total else 0
        assert AR.gate_residual(cells(r), nxt, fixed[int(kinds[r])], P) == 0, (r, int(kinds[r]))
