#!/usr/bin/env python3
"""Time the device-side flatten (emit_kernel) on a pow trace: per-launch ms from the C ABI's dispatch-stamped events,
stream bytes written per second and read + written bytes per second.
usage: emit_timing.py [batch] [workload: rsa2048 | rsa4096w32 | rsa3072] [flags]"""
import os, sys, random
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
import halo2_rsa_amd as H
from halo2_rsa_amd import _lib
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
wl = sys.argv[2] if len(sys.argv) > 2 else "rsa2048"
flags = int(sys.argv[3]) if len(sys.argv) > 3 else 0
w, bits = {"rsa2048": (64, 2048), "rsa4096w32": (32, 4096), "rsa3072": (64, 3072), "rsa1024": (64, 1024)}[wl]
chip = H.BigIntChip(w, bits)
rng = random.Random(5)
N = [rng.getrandbits(bits) | (1 << (bits - 1)) | 1 for _ in range(B)]
X = [rng.randrange(n) for n in N]
res = chip.pow_mod_fixed_exp(chip.assign_integer(X), 65537, chip.assign_integer(N))
torch.cuda.synchronize()
tr = res.trace
sb = tr.stream_bytes_ex(flags)
out = torch.zeros((B, sb), dtype=torch.uint8, device="cuda")
tr.emit_stream(flags, out=out); torch.cuda.synchronize()
_lib.profile_enable(64)
for _ in range(10):
    tr.emit_stream(flags, out=out)
torch.cuda.synchronize()
ms = _lib.profile_read(_lib.KERNEL_EMIT)
_lib.profile_enable(0)
avg = sum(ms) / len(ms)
wr = B * sb
rd = B * tr.num_mul_mods * chip.layout.stream_bytes   # algorithmic bytes of the records read
print("emit_kernel %s batch %d flags %d: %.3f ms per launch (min %.3f)  stream written %.2f TB/s, read+written %.2f TB/s  (%d B/element)"
      % (wl, B, flags, avg, min(ms), wr / avg / 1e9, (wr + rd) / avg / 1e9, sb))
# the same with the stream written into a region of the placement-aware arena (where the output lies decides the store rate)
if wl == "rsa2048" and flags == 0:
    arena = H.TraceArena.for_pow(chip, 65537, B, regions=1, candidates=12)
    out_a = arena.regions[0][:B * sb].view(B, sb)
    tr.emit_stream(flags, out=out_a); torch.cuda.synchronize()
    _lib.profile_enable(64)
    for _ in range(10):
        tr.emit_stream(flags, out=out_a)
    torch.cuda.synchronize()
    ms_a = _lib.profile_read(_lib.KERNEL_EMIT)
    _lib.profile_enable(0)
    avg_a = sum(ms_a) / len(ms_a)
    print("emit_kernel %s batch %d flags %d, stream written into an arena region: %.3f ms per launch  read+written %.2f TB/s"
          % (wl, B, flags, avg_a, (wr + rd) / avg_a / 1e9))
    del out_a
    arena.close()
# the 5-column advice image of the same trace (advice_kernel): 160-byte rows of field elements
if len(sys.argv) <= 4 or sys.argv[4] != "noadvice":
    img = res.emit_advice(); torch.cuda.synchronize()
    nbytes = img.numel()
    del img
    _lib.profile_enable(16)
    for _ in range(3):
        img = res.emit_advice(); del img
    torch.cuda.synchronize()
    ms = _lib.profile_read(_lib.KERNEL_EMIT)
    _lib.profile_enable(0)
    avg = sum(ms) / len(ms)
    print("advice_kernel %s batch %d: %.3f ms per launch (min %.3f)  image written %.2f TB/s  (%d rows x 160 B per mul_mod, %.2f MB per element)"
          % (wl, B, avg, min(ms), nbytes / avg / 1e9, int(_lib.lib().h2r_advice_rows(chip._ctx)), nbytes / B / 1e6))

    # the same image written into a region of the placement-aware arena (one region of ten trace regions' size, the fastest of 6 candidates)
    if wl == "rsa2048":
        try:
            arena = H.TraceArena.for_pow(chip, 65537, 10 * B, regions=1, candidates=6)
            reg = arena.regions[0]
            img = res.emit_advice(out=reg); torch.cuda.synchronize()
            _lib.profile_enable(16)
            for _ in range(3):
                res.emit_advice(out=reg)
            torch.cuda.synchronize()
            ms = _lib.profile_read(_lib.KERNEL_EMIT)
            _lib.profile_enable(0)
            avg = sum(ms) / len(ms)
            print("advice_kernel %s batch %d, image written into an arena region (kept %.3f ms, candidates %s): %.3f ms per launch (min %.3f)  image written %.2f TB/s"
                  % (wl, B, arena.region_ms[0], ["%.2f" % t for t in arena.measurements_ms], avg, min(ms), nbytes / avg / 1e9))
            del img, reg
            arena.close()
        except Exception as ex:
            print("advice_kernel into an arena region: skipped (%s)" % str(ex)[:200])
