"""GPU: halo2's lookup argument for the range checks (h2r_lookup_*) against the Python restatement of RangeChip's placement
and of halo2's permute_expression_pair (tests/advice_ref.py) run on the ORACLE's cells."""
import random

import numpy as np
import pytest
import torch

import advice_ref as AR
from oracle_lib import Oracle
from test_maingate_image_ref import FIELDS

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def H():
    import halo2_rsa_amd as H
    return H


def _circuit_reference(o, chip_w, L, P, cfg, x, n, e, usable):
    """The oracle's cells of one circuit = assign_integer(x), assign_integer(n) (big_integer/chip.rs:71-76) followed by every
    mul_mod of pow_mod_fixed_exp(x, e, n), as main-gate rows; returns per-argument (tag, value) inputs over `usable` rows."""
    w = chip_w
    rows, kinds = [], []
    pre = AR.Image(w, L, P)
    for v in (x, n):
        for k in range(L):
            limb = (v >> (w * k)) & ((1 << w) - 1)
            pre.range_assign(limb, [(limb >> ((w // 8) * t)) & ((1 << (w // 8)) - 1) for t in range(8)], w // 8, AR.ROW_RANGE_LIMB)
    rows += pre.rows
    kinds += pre.kinds
    # the mul_mods in the reference's call order (chip.rs:731-740), each from the oracle's own stream
    nl = [int(t) for t in o.limbs(n)]
    acc, sq = 1, x
    bits = [(e >> i) & 1 for i in range(e.bit_length())]
    calls = []
    for bit in bits:
        cur = sq
        calls.append((cur, cur))
        sq = cur * cur % n
        if bit:
            calls.append((acc, cur))
            acc = acc * cur % n
    for (a, b) in calls:
        rc, r, st = o.mul_mod(o.limbs(a), o.limbs(b), o.limbs(n))
        assert rc == 0
        im = AR.mul_mod_image(o.p, [int(t) for t in o.limbs(a)], [int(t) for t in o.limbs(b)], nl, st, P)
        rows += im.rows
        kinds += im.kinds
    fixed = [AR.fixed_row(k, w, L, o.p.carry_bits, o.p.carry_sub_bits, o.p.carry_nsub, cfg) for k in kinds]
    assert len(rows) <= usable
    return AR.lookup_inputs(rows, fixed, usable), len(calls), acc


@pytest.mark.parametrize("w,L,e,field,rsa", [(64, 32, 5, "bn254_fr", True), (32, 8, 0b1011, "pasta_fp", False),
                                             (64, 4, 65537, "bn254_fq", False), (32, 128, 3, "pasta_fq", False)])
def test_lookup_hist_and_permuted_columns(H, w, L, e, field, rsa):
    from halo2_rsa_amd import _lib
    P = FIELDS[field]
    chip = H.BigIntChip(w, w * L, field=field)
    o = Oracle(w, L)
    rng = random.Random(31 * w + L)
    bits = w * L
    batch = 3
    N = [rng.getrandbits(bits) | (1 << (bits - 1)) | 1 for _ in range(batch)]
    X = [rng.randrange(n) for n in N]
    x_dev, n_dev = chip.assign_integer(X), chip.assign_integer(N)
    res = chip.pow_mod_fixed_exp(x_dev, e, n_dev)
    la = H.LookupArgument(chip, rsa_chip=rsa)
    cfg = AR.LookupConfig(AR.range_lens(w, L, rsa=rsa))
    assert la.table_image() == cfg.table()
    hist = la.new_hist(batch)
    la.hist_values(x_dev.limbs_dev, w, w // 8, hist)          # assign_integer(x), assign_integer(n): chip.rs:71-76
    la.hist_values(n_dev.limbs_dev, w, w // 8, hist)
    la.hist_records(res.trace, hist, res.status)
    torch.cuda.synchronize()
    assert not res.status.cpu().numpy().any()
    rows_per_circuit = 4 * L + res.trace.num_mul_mods * (4 * L + 2 * (2 * L - 1 + L * L) + L + 4 + (2 * L - 2) * (23 + (o.p.carry_nsub + 3) // 4) + 23)
    k = max(rows_per_circuit + 6, cfg.n_rows + 6).bit_length()
    usable = (1 << k) - 6          # blinding_factors + 1 = 6 rows of a main-gate circuit are not usable
    thetas = [rng.randrange(P), 1, P - 1][:batch]            # theta = 1 makes table rows collide ((1,1) and (2,0) both compress to 2)
    a_perm, s_perm, status = la.permuted_columns(hist, thetas, usable)
    torch.cuda.synchronize()
    assert not status.cpu().numpy().any()
    hist_h = hist.cpu().numpy()
    row_of = {pair: i for i, pair in enumerate(cfg.table())}
    for b in range(batch):
        inputs, n_calls, acc = _circuit_reference(o, w, L, P, cfg, X[b], N[b], e, usable)
        assert n_calls == res.trace.num_mul_mods and res.value.to_big_uint()[b] == acc
        tcol = AR.table_column(cfg, thetas[b], usable, P)
        for k_arg, name in enumerate(AR.ARGS):
            want_hist = np.zeros(cfg.n_rows, dtype=np.int64)
            for pair in inputs[name]:
                if pair != (0, 0):
                    want_hist[row_of[pair]] += 1
            assert np.array_equal(hist_h[b, k_arg], want_hist), (b, name)
            A = AR.compress(inputs[name], thetas[b], P)
            a_ref, s_ref = AR.permute_expression_pair(A, tcol)
            got_a = a_perm[b, k_arg].cpu().numpy().tobytes()
            got_s = s_perm[b, k_arg].cpu().numpy().tobytes()
            assert got_a == b"".join(v.to_bytes(32, "little") for v in a_ref), (b, name, "A'")
            assert got_s == b"".join(v.to_bytes(32, "little") for v in s_ref), (b, name, "S'")


def test_lookup_with_a_custom_tag_map(H):
    """h2r_lookup_config_custom: a maingate revision that tags every bit length with the bit length itself (instead of 1, 2, ...):
    table image, multiplicities and permuted columns against the Python restatement configured the same way."""
    w, L, P = 64, 8, FIELDS["bn254_fr"]
    chip = H.BigIntChip(w, w * L)
    o = Oracle(w, L)
    lens = sorted(set(b for b in AR.range_lens(w, L) if b))
    la = H.LookupArgument(chip, bit_lens=lens, tags=lens)
    cfg = AR.LookupConfig(lens, tags=lens)
    assert la.table_image() == cfg.table()
    rng = random.Random(99)
    n = rng.getrandbits(w * L) | (1 << (w * L - 1)) | 1
    x = rng.randrange(n)
    x_dev, n_dev = chip.assign_integer([x]), chip.assign_integer([n])
    res = chip.pow_mod_fixed_exp(x_dev, 3, n_dev)
    hist = la.new_hist(1)
    la.hist_values(x_dev.limbs_dev, w, 8, hist)
    la.hist_values(n_dev.limbs_dev, w, 8, hist)
    la.hist_records(res.trace, hist, res.status)
    usable = (1 << 12) - 6
    theta = rng.randrange(P)
    a_perm, s_perm, status = la.permuted_columns(hist, [theta], usable)
    torch.cuda.synchronize()
    assert status.cpu().tolist() == [0]
    inputs, n_calls, acc = _circuit_reference(o, w, L, P, cfg, x, n, 3, usable)
    tcol = AR.table_column(cfg, theta, usable, P)
    for k_arg, name in enumerate(AR.ARGS):
        a_ref, s_ref = AR.permute_expression_pair(AR.compress(inputs[name], theta, P), tcol)
        assert a_perm[0, k_arg].cpu().numpy().tobytes() == b"".join(v.to_bytes(32, "little") for v in a_ref), name
        assert s_perm[0, k_arg].cpu().numpy().tobytes() == b"".join(v.to_bytes(32, "little") for v in s_ref), name
    # a challenge that is not a canonical field element is refused (status), the columns are left alone
    a2, s2, st2 = la.permuted_columns(hist, [P + 5], usable, out=(torch.zeros_like(a_perm), torch.zeros_like(s_perm)))
    torch.cuda.synchronize()
    assert st2.cpu().tolist() == [H.H2R_E_SHAPE] and not a2.any() and not s2.any()


def test_lookup_rejects_what_does_not_fit(H):
    from halo2_rsa_amd import _lib
    chip = H.BigIntChip(64, 2048)
    la = H.LookupArgument(chip)
    hist = la.new_hist(2)
    hist[1, 0, 5] = 1000
    a, s, status = la.permuted_columns(hist, [7, 7], 512)     # 339 table rows fit 512 usable rows; 1,000 inputs do not
    torch.cuda.synchronize()
    assert status.cpu().tolist() == [0, _lib.H2R_E_SHAPE]
    with pytest.raises(_lib.H2RError):
        la.permuted_columns(hist, [7, 7], 300)                # fewer usable rows than table rows
    res = chip.mul_mod(chip.assign_integer([1]), chip.assign_integer([1]), chip.assign_integer([3]))
    no6 = H.LookupArgument(chip, bit_lens=[8, 1], tags=[1, 2])          # a table without the 6-bit overflow length of the carries
    with pytest.raises(_lib.H2RError):
        no6.hist_records(res.trace, no6.new_hist(1))                     # RangeChip::assign has no table for that bit length


def test_lookup_full_size_invariants(H):
    """BASELINE config 2's size: 1,024 circuits of k = 17 (131,066 usable rows), every argument -- the lookup argument's own
    invariants checked on the DEVICE for every row of every column: A' sorted ascending and a permutation of the inputs
    (multiset equality through the run lengths = the multiplicities), S' a permutation of the table column, and on every row
    A'[i] == S'[i] or A'[i] == A'[i-1]."""
    chip = H.BigIntChip(64, 2048)
    rng = random.Random(2048)
    B = 1024
    N = [rng.getrandbits(2048) | (1 << 2047) | 1 for _ in range(B)]
    X = [rng.randrange(n) for n in N]
    x_dev, n_dev = chip.assign_integer(X), chip.assign_integer(N)
    res = chip.pow_mod_fixed_exp(x_dev, 65537, n_dev)
    la = H.LookupArgument(chip)
    hist = la.new_hist(B)
    la.hist_values(x_dev.limbs_dev, 64, 8, hist)
    la.hist_values(n_dev.limbs_dev, 64, 8, hist)
    la.hist_records(res.trace, hist, res.status)
    usable = (1 << 17) - 6
    P = FIELDS["bn254_fr"]
    thetas = [rng.randrange(P) for _ in range(B)]
    CH = 128                                   # circuits per call: 128 x 5 x 2 x 4.2 MB = 5.4 GB of columns at a time
    tab = la.table_image()
    for c0 in range(0, B, CH):
        a_perm, s_perm, status = la.permuted_columns(hist[c0:c0 + CH].contiguous(), thetas[c0:c0 + CH], usable)
        assert not status.cpu().numpy().any()
        A = a_perm.view(torch.int64).view(CH, 5, usable, 4)      # little-endian words; values < 2^255: signed compare of the top word is fine
        S = s_perm.view(torch.int64).view(CH, 5, usable, 4)
        eq_as = (A == S).all(-1)
        eq_prev = torch.zeros_like(eq_as)
        eq_prev[:, :, 1:] = (A[:, :, 1:] == A[:, :, :-1]).all(-1)
        assert bool((eq_as | eq_prev).all())                     # the adjacency rule of the lookup argument
        # ascending: compare as (w3, w2, w1, w0) tuples; words are < 2^63 only for w3, so compare the others as unsigned via xor of the sign bit
        def key(T):
            return [T[..., 3], T[..., 2] ^ (-2 ** 63), T[..., 1] ^ (-2 ** 63), T[..., 0] ^ (-2 ** 63)]
        ka, kb = key(A[:, :, :-1]), key(A[:, :, 1:])
        le = torch.ones_like(ka[0], dtype=torch.bool)
        decided = torch.zeros_like(le)
        for u, v in zip(ka, kb):
            le = torch.where(~decided & (u > v), torch.zeros_like(le), le)
            decided = decided | (u != v)
        assert bool(le.all())
        # multiset equality: the number of run heads of A' = distinct values with inputs, and every circuit has the same number of
        # zero inputs as rows without a lookup; S' sums (as integers mod 2^64 per word) equal the table column's sums
        heads = (~eq_prev).sum(-1)
        nz = (hist[c0:c0 + CH] > 0).sum(-1) + 1                 # + the (0, 0) run
        assert torch.equal(heads.to(torch.int64), nz.to(torch.int64))
        for b in (0, CH - 1):                                    # two circuits per chunk: exact multiset check on the host
            th = thetas[c0 + b]
            tcol = sorted([(t * th + v) % P for (t, v) in tab] + [0] * (usable - len(tab)))
            for k_arg in (0, 4):
                got = sorted(int.from_bytes(bytes(r), "little") for r in s_perm[b, k_arg].cpu().numpy())
                assert got == tcol
                h = hist[c0 + b, k_arg].cpu().tolist()
                want_a = sorted([0] * (usable - sum(h)) + [((tab[i][0] * th + tab[i][1]) % P) for i, m in enumerate(h) for _ in range(m)])
                got_a = [int.from_bytes(bytes(r), "little") for r in a_perm[b, k_arg].cpu().numpy()]
                assert got_a == want_a
        del a_perm, s_perm, A, S


def _in_field_range_assigns(stream, L, w):
    """(value, [8 sub-limbs]) of every RangeChip::assign inside the oracle's is_in_field stream (big_integer/chip.rs:908-919 ->
    sub :310-373 -> add :245-297 (c and carry per limb, :279-282) and sub_unchecked :1286-1318 (difference limbs, :1307-1308))."""
    LB = w // 8
    SB = RA = LB + 8
    st = bytes(stream)
    pos = 0
    out = []

    def ra():
        nonlocal pos
        v = int.from_bytes(st[pos:pos + LB], "little")
        subs = list(st[pos + LB:pos + LB + 8])
        assert v == sum(s << ((w // 8) * t) for t, s in enumerate(subs))
        out.append((v, subs))
        pos += RA

    def add(n):
        nonlocal pos
        for _ in range(n):
            pos += 2 * SB
            ra(); ra()
            pos += SB

    def eq(n):
        nonlocal pos
        pos += 2 * n

    def subu(n1):
        for _ in range(n1):
            ra()
        add(n1); eq(n1 + 1)

    def sub(nA, nB):
        nonlocal pos
        m = max(nA, nB)
        n1 = m + 1
        add(m); subu(n1)
        pos += 2 + n1 * LB + m * LB
        subu(n1)
    sub(L, L); eq(L)
    pos += 2
    assert pos == len(st)
    return out


@pytest.mark.parametrize("w,L", [(64, 32), (32, 16)])
def test_lookup_hist_of_the_in_field_witness(H, w, L):
    """h2r_lookup_hist_fresh_op on the assert_in_field witness of modpow_public_key: the multiplicities of its 8 L + 6 range
    assigns per argument against a count over the ORACLE's in-field stream."""
    chip = H.BigIntChip(w, w * L)
    o = Oracle(w, L)
    rng = random.Random(w * L)
    bits = w * L
    N = [rng.getrandbits(bits) | (1 << (bits - 1)) | 1 for _ in range(4)]
    X = [rng.randrange(n) for n in N]
    X[3] = N[3] + 1 if N[3] + 1 < (1 << bits) else N[3]          # not in the field: the witness is still written
    res = chip.pow_mod_fixed_exp(chip.assign_integer(X), 3, chip.assign_integer(N), check_in_field=True, want_trace=True)
    la = H.LookupArgument(chip, rsa_chip=False)
    cfg = AR.LookupConfig(AR.range_lens(w, L))
    hist = la.new_hist(4)
    es = chip.in_field_layout()[0]
    la.hist_fresh_op("is_in_field", res.in_field.buf, es, 4, hist)
    torch.cuda.synchronize()
    got = hist.cpu().numpy()
    off = cfg.row_off[w // 8]
    for b in range(4):
        rc, lt, st = o.assert_in_field(o.limbs(X[b]), o.limbs(N[b]))
        ras = _in_field_range_assigns(st, L, w)
        assert len(ras) == 8 * L + 6            # add(L): 2L, two sub_unchecked of L + 1 limbs: 2 (L + 1) + 2 * 2 (L + 1)
        want = np.zeros((5, cfg.n_rows), dtype=np.int64)
        for (_, subs) in ras:
            for t, sv in enumerate(subs):
                want[t if t < 4 else 7 - t, off + sv] += 1
        assert np.array_equal(got[b], want), b


def test_lookup_hist_of_a_whole_verify_element(H, golden):
    """h2r_lookup_hist_verify + h2r_lookup_hist_values(sig), (n): the multiplicities of EVERY lookup of one
    RSAChip::verify_pkcs1v15_signature circuit (assign_signature, assign_public_key, assert_in_field, pow_mod_fixed_exp, the
    encoded-message check's 4-bit range rows) equal a count over the Python restatement's rows built from the ORACLE's streams;
    the permuted columns of that circuit satisfy the lookup argument's invariants."""
    import ctypes
    import os
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
    import pyref as R
    from halo2_rsa_amd._lib import lib
    P = R.FIELD_MODULI["bn254_fr"]
    e = 5                                                        # 4 mul_mods: keeps the Python image small
    k = golden["rsa_kats"][0]
    rng = random.Random(31)
    ns = [int(k["n"]), rng.getrandbits(2048) | (1 << 2047) | 1]
    sigs = [int(k["sig"]), rng.randrange(ns[1])]
    hashed = [int(k["hashed"]), rng.getrandbits(256)]
    rsa = H.RSAChip(2048, 5)
    chip = rsa.bigint_chip()
    pk = rsa.assign_public_key(H.RSAPublicKey(H.UnassignedInteger.from_ints(ns, 32, 64), H.Fix(e)))
    sg = rsa.assign_signature(H.RSASignature(H.UnassignedInteger.from_ints(sigs, 32, 64)))
    res = rsa.verify_pkcs1v15_signature(pk, hashed, sg)
    la = H.LookupArgument(chip, rsa_chip=True)
    hist = la.new_hist(2)
    sig_d, n_d, _ = res.inputs
    la.hist_values(sig_d.limbs_dev, 64, 8, hist)
    la.hist_values(n_d.limbs_dev, 64, 8, hist)
    la.hist_verify(res, hist)
    torch.cuda.synchronize()
    assert res.status.cpu().tolist() == [0, 0]
    got = hist.cpu().numpy()
    cfg = AR.LookupConfig(AR.range_lens(64, 32, rsa=True))
    o = Oracle(64, 32)
    usable = 1 << 15
    for b in range(2):
        inputs, n_calls, acc = _circuit_reference(o, 64, 32, P, cfg, sigs[b], ns[b], e, usable)    # assign_integer x 2 + the mul_mods
        rc, lt, s_if = o.assert_in_field(o.limbs(sigs[b]), o.limbs(ns[b]))
        im_if = AR.in_field_image(o.p, o.limbs(sigs[b]), o.limbs(ns[b]), s_if, P)
        powed = o.limbs(pow(sigs[b], e, ns[b]))
        rc, valid, s_em = o.pkcs1v15_em_check(powed, o.limbs(hashed[b], 4))
        im_em, _ = AR.em_image(o.p, powed, o.limbs(hashed[b], 4), s_em, P)
        rows = im_if.rows + im_em.rows
        fixed = [AR.fixed_row(kk, 64, 32, o.p.carry_bits, o.p.carry_sub_bits, o.p.carry_nsub, cfg) for kk in im_if.kinds + im_em.kinds]
        extra = AR.lookup_inputs(rows, fixed, len(rows))
        want = np.zeros((5, cfg.n_rows), dtype=np.int64)
        row_of = {tv: i for i, tv in enumerate(cfg.table())}
        for a, name in enumerate(AR.ARGS):
            for (tag, val) in list(inputs[name]) + list(extra[name]):
                if tag:
                    want[a, row_of[(tag, val)]] += 1
        g = got[b].astype(np.int64).copy()
        g[:, 0] = 0                                              # row 0 = (0, 0): the rows with the lookup off, a function of usable_rows
        want[:, 0] = 0
        assert np.array_equal(g, want), (b, np.argwhere(g != want)[:5])
        assert want[:, cfg.row_off[4]:cfg.row_off[4] + 16].sum() == 16      # the EM check's sixteen 4-bit sub-limbs
    # strided form: the same two halves counted from powed limb 6 viewed as two 32-bit words
    h2 = la.new_hist(2)
    assert lib().h2r_lookup_hist_values_strided(chip._ctx, ctypes.byref(la.cfg), res.powed.limbs_dev.data_ptr() + 48, 4, 2, 2, 32 * 8, 4, 32, 4,
                                                None, h2.data_ptr(), chip._stream()) == 0
    h3 = la.new_hist(2)
    assert lib().h2r_lookup_hist_values_strided(chip._ctx, ctypes.byref(la.cfg), res.trace.data_ptr() + res.layout.off_em + 12, 4, 2, 2,
                                                res.layout.elem_stride, 12, 32, 4, res.status.data_ptr(), h3.data_ptr(), chip._stream()) == 0
    torch.cuda.synchronize()
    assert torch.equal(h2, h3) and int(h2.sum()) == 2 * 16
    assert lib().h2r_lookup_hist_values_strided(chip._ctx, ctypes.byref(la.cfg), res.trace.data_ptr() + 2, 4, 2, 2, 256, 12, 32, 4, None,
                                                h3.data_ptr(), chip._stream()) == H.H2R_E_SHAPE
