"""cells_kernel (H2R_ADVICE_DIRECT) against advice_kernel: byte comparison of the two images for several shapes, then timing of
both at BASELINE config 2's size.  Usage: python tools/cells_check.py [--time] [--batch N]"""
import argparse
import ctypes
import os
import random
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import halo2_rsa_amd as H  # noqa: E402
from halo2_rsa_amd._lib import lib  # noqa: E402


def rand_modulus(rng, bits, odd=True):
    n = rng.getrandbits(bits) | (1 << (bits - 1))
    return n | 1 if odd else n & ~1


def first_diff(got, want, rows, kinds):
    g, w = got.reshape(-1, rows, 160), want.reshape(-1, rows, 160)
    bad = np.argwhere(g != w)
    if not len(bad):
        return None
    e, r, b = (int(v) for v in bad[0])
    nbad_rows = len({(int(x[0]), int(x[1])) for x in bad[:100000]})
    return "elem %d row %d (kind %d) cell %d: got %s want %s; %d differing rows (of first 100k bytes)" % (
        e, r, int(kinds[r]), b // 32, g[e, r, 32 * (b // 32):32 * (b // 32) + 32].tobytes()[::-1].hex(),
        w[e, r, 32 * (b // 32):32 * (b // 32) + 32].tobytes()[::-1].hex(), nbad_rows)


def check_shape(w, L, field, batch=5):
    chip = H.BigIntChip(w, w * L, field=field)
    rng = random.Random(w * 1000 + L)
    N = [rand_modulus(rng, w * L, odd=(i != 1)) for i in range(batch)]
    A = [rng.randrange(n) for n in N]
    B = [rng.randrange(n) for n in N]
    A[2] = N[2] - 1
    B[2] = N[2] - 1
    A[3] = 0
    rows = int(lib().h2r_advice_rows(chip._ctx))
    kinds = np.zeros(rows, dtype=np.uint8)
    lib().h2r_advice_row_kinds(chip._ctx, kinds.ctypes.data)
    res = chip.mul_mod(chip.assign_integer(A), chip.assign_integer(B), chip.assign_integer(N))
    want = res.emit_advice().cpu().numpy()
    got = res.emit_advice(direct=True).cpu().numpy()
    d = first_diff(got, want, rows, kinds)
    print("mul_mod w=%d L=%d %s: %s" % (w, L, field, "OK" if d is None else "DIFF " + d), flush=True)
    ok = d is None
    for e in (0b1011, 65537):
        pres = chip.pow_mod_fixed_exp(chip.assign_integer(A), e, chip.assign_integer(N))
        want = pres.emit_advice().cpu().numpy()
        got = pres.emit_advice(direct=True).cpu().numpy()
        T = pres.trace.num_mul_mods
        same_pre = np.array_equal(got[:, :320], want[:, :320])
        d = first_diff(got[:, 320:], want[:, 320:], rows, kinds)
        print("pow e=%d w=%d L=%d: pre rows %s, records %s" % (e, w, L, "OK" if same_pre else "DIFF", "OK" if d is None else "DIFF " + d), flush=True)
        ok = ok and d is None and same_pre
        # without records at all
        p2 = chip.pow_mod_fixed_exp(chip.assign_integer(A), e, chip.assign_integer(N), want_trace=False,
                                    workspace=torch.empty(chip.workspace_bytes(batch, T), dtype=torch.uint8, device="cuda"))
        got2 = p2.emit_advice(direct=True).cpu().numpy()
        same = np.array_equal(got2, want)
        print("   no-record call: %s" % ("OK" if same else "DIFF"), flush=True)
        ok = ok and same
    return ok


def timing(batch, steps=6):
    chip = H.BigIntChip(64, 2048)
    rng = random.Random(7)
    N = [rand_modulus(rng, 2048) for _ in range(batch)]
    X = [rng.randrange(n) for n in N]
    pres = chip.pow_mod_fixed_exp(chip.assign_integer(X), 65537, chip.assign_integer(N))
    torch.cuda.synchronize()
    nrows = int(lib().h2r_pow_advice_rows(chip._ctx, ctypes.byref(pres.trace.pow_layout)))
    out = torch.empty(batch * nrows * 160, dtype=torch.uint8, device="cuda")
    for direct in (False, True, False, True):
        for _ in range(2):
            pres.emit_advice(out=out, direct=direct)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            pres.emit_advice(out=out, direct=direct)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / steps
        print("batch %d %s: %.3f ms = %.2f TB/s written" % (batch, "cells_kernel " if direct else "advice_kernel", dt * 1e3, batch * nrows * 160 / dt / 1e12), flush=True)


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--time", action="store_true")
    ap.add_argument("--batch", type=int, default=1024)
    ap.add_argument("--skip-check", action="store_true")
    args = ap.parse_args()
    ok = True
    if not args.skip_check:
        for (w, L, field) in [(64, 32, "bn254_fr"), (64, 16, "bn254_fq"), (32, 128, "pasta_fp"), (64, 12, "bn254_fq"), (64, 48, "pasta_fq"), (32, 8, "bn254_fr"), (64, 64, "bn254_fr"), (64, 4, "pasta_fq")]:
            try:
                ok = check_shape(w, L, field) and ok
            except Exception as ex:  # keep going: one run should tell as much as possible
                print("w=%d L=%d: EXCEPTION %r" % (w, L, ex), flush=True)
                ok = False
    if args.time:
        timing(args.batch)
    sys.exit(0 if ok else 1)
