#!/usr/bin/env python3
"""Kernel times (C ABI dispatch-stamped events) of the exports OFF the modpow hot path at batch 1024: the Fresh-integer
family, BigIntChip::mul, is_equal_muled, refresh, the in-field / encoded-message kernel, the device-side flatten and the
in-place audit.  usage: offpath_timing.py [batch] [w bits]   (default 1024, 64 2048)"""
import os, sys, random
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
import halo2_rsa_amd as H
from halo2_rsa_amd import _lib
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
w, bits = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (64, 2048)
chip = H.BigIntChip(w, bits)
rng = random.Random(9)
N = [rng.getrandbits(bits) | (1 << (bits - 1)) | 1 for _ in range(B)]
A = [rng.randrange(n) for n in N]
Bv = [rng.randrange(n) for n in N]
an, aa, ab = chip.assign_integer(N), chip.assign_integer(A), chip.assign_integer(Bv)


def timed(name, fn, kernel, reps=5, bytes_per_elem=None):
    fn(); torch.cuda.synchronize()
    _lib.profile_enable(4 * reps + 8)
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    ms = _lib.profile_read(kernel)
    _lib.profile_enable(0)
    avg = sum(ms) / len(ms) if ms else float("nan")
    extra = "  %.2f TB/s of flat stream" % (B * bytes_per_elem / avg / 1e9) if bytes_per_elem else ""
    print("%-28s %8.4f ms per batch of %d (min %.4f, %d launches)%s" % (name, avg, B, min(ms) if ms else 0, len(ms), extra), flush=True)


for op in ("add", "sub", "add_mod", "sub_mod", "is_zero", "is_equal_fresh", "is_less_than", "is_in_field"):
    f = getattr(chip, op)
    args = (aa, ab, an) if op in ("add_mod", "sub_mod") else ((aa,) if op == "is_zero" else (aa, an))
    r = f(*args)
    timed("fresh " + op, lambda: f(*args), _lib.KERNEL_AUX, bytes_per_elem=r.stream_bytes)
mul = chip.mul(aa, ab)
timed("mul (a*b columns)", lambda: chip.mul(aa, ab), _lib.KERNEL_TRACE, bytes_per_elem=int(_lib.lib().h2r_mul_stream_bytes(chip._ctx)))
timed("is_equal_muled", lambda: chip.is_equal_muled(mul, mul), _lib.KERNEL_TRACE, bytes_per_elem=int(_lib.lib().h2r_is_equal_muled_stream_bytes(chip._ctx)))
timed("refresh", lambda: chip.refresh(mul), _lib.KERNEL_AUX, bytes_per_elem=int(_lib.lib().h2r_refresh_stream_bytes(chip._ctx)))
mm = chip.mul_mod(aa, ab, an)
timed("mul_mod record kernel", lambda: chip.mul_mod(aa, ab, an), _lib.KERNEL_TRACE, bytes_per_elem=chip.layout.stream_bytes)
timed("mul_mod emit_stream", lambda: mm.trace.emit_stream(), _lib.KERNEL_EMIT, bytes_per_elem=chip.layout.stream_bytes)
timed("mul_mod emit_advice", lambda: mm.emit_advice(), _lib.KERNEL_EMIT, bytes_per_elem=int(_lib.lib().h2r_advice_rows(chip._ctx)) * 160)
