"""h2r_lookup_hist_advice on BASELINE config 2's image (1,024 RSA-2048 modpow_public_key elements, 12.4 GB): ms per call, next to the
record-based count of the same elements (h2r_lookup_hist_records + _fresh_op)."""
import ctypes, os, random, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import halo2_rsa_amd as H
from halo2_rsa_amd import _lib
from halo2_rsa_amd._lib import lib

B = 1024
for kw in (dict(), dict(columns=True, montgomery=True)):
    chip = H.BigIntChip(64, 2048, **kw)
    la = H.LookupArgument(chip, rsa_chip=False)
    rng = random.Random(1)
    N = [rng.getrandbits(2048) | (1 << 2047) | 1 for _ in range(B)]
    X = [rng.randrange(n) for n in N]
    res = chip.pow_mod_fixed_exp(chip.assign_integer(X), 65537, chip.assign_integer(N), check_in_field=True)
    pl = res.trace.pow_layout
    k_if = chip.fresh_op_row_kinds(_lib.FRESH_OPS.index("is_in_field"), assert_one=True)
    k_pow = np.zeros(int(lib().h2r_pow_advice_rows(chip._ctx, ctypes.byref(pl))), dtype=np.uint8)
    lib().h2r_pow_row_kinds(chip._ctx, ctypes.byref(pl), k_pow.ctypes.data)
    kd = torch.from_numpy(np.concatenate([k_if, k_pow])).cuda()
    img = res.emit_modpow_advice()
    want = la.hist_records(res.trace, la.new_hist(B), status=res.status)
    la.hist_fresh_op("is_in_field", res.in_field.buf, res.in_field.elem_stride, B, want)

    def t(fn, n=5):
        fn(); torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(n):
            fn()
        b.record(); torch.cuda.synchronize()
        return a.elapsed_time(b) / n
    h = la.new_hist(B)
    ms_img = t(lambda: la.hist_advice(kd, img, B, h, status=res.status))
    h2 = la.new_hist(B)
    ms_rec = t(lambda: la.hist_records(res.trace, h2, status=res.status))
    got = la.hist_advice(kd, img, B, la.new_hist(B), status=res.status)
    torch.cuda.synchronize()
    print("%-40s image %.1f GB (its lookup rows, ~8 %%, are read): hist_advice %.3f ms, hist_records %.3f ms; equal: %s" % (kw or "canonical row-major", img.numel() / 1e9, ms_img,
                                                                                                                    ms_rec, bool(torch.equal(got, want))))
    del res, img
    torch.cuda.empty_cache()
