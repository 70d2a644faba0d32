#!/usr/bin/env python3
"""Record-kernel time per trace BUFFER: N buffer sets allocated back to back, pipelined batch-1024 RSA-2048 calls rotating
through them, mean kernel time per buffer (the speed of the record kernel depends on where its output buffer lies).
usage: buffer_speed_probe.py [buffers] [calls]    (H2R_LIB / H2R_TRACE_MAP select build and mapping)"""
import os, sys, random
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
import halo2_rsa_amd as H
from halo2_rsa_amd import _lib
nset = int(sys.argv[1]) if len(sys.argv) > 1 else 6
CALLS = int(sys.argv[2]) if len(sys.argv) > 2 else 120
chip = H.BigIntChip(64, 2048); pl = chip.pow_fixed_layout(65537)
rng = random.Random(1); B = 1024
N = [rng.getrandbits(2048) | (1 << 2047) | 1 for _ in range(B)]; X = [rng.randrange(n) for n in N]
n, x = chip.assign_integer(N), chip.assign_integer(X)
sets = [dict(trace=torch.zeros(B * pl.elem_stride, dtype=torch.uint8, device="cuda"), ws=torch.zeros(chip.workspace_bytes(B, pl.num_mul_mods), dtype=torch.uint8, device="cuda"),
             out=torch.zeros((B, 32), dtype=torch.int64, device="cuda"), status=torch.zeros(B, dtype=torch.uint8, device="cuda")) for _ in range(nset)]
pipe = H.Pipeline(chip, depth=2, side_streams=1)
for k in range(2 * nset):
    s = sets[k % nset]; pipe.modpow_public_key(x, 65537, n, s["trace"], s["ws"], s["out"], s["status"])
pipe.join(); torch.cuda.synchronize()
_lib.profile_enable(4 * CALLS)
for k in range(CALLS):
    s = sets[k % nset]; pipe.modpow_public_key(x, 65537, n, s["trace"], s["ws"], s["out"], s["status"])
pipe.join(); torch.cuda.synchronize()
ms = _lib.profile_read(_lib.KERNEL_TRACE); _lib.profile_enable(0)
print("mapping", os.environ.get("H2R_TRACE_MAP", "default"))
for j in range(nset):
    v = ms[j::nset]
    print("buffer %d at 0x%x: record kernel mean %.4f ms (min %.4f max %.4f) %.0f GB/s" % (j, sets[j]["trace"].data_ptr(), sum(v) / len(v), min(v), max(v), B * 19 * 64338 / (sum(v) / len(v)) / 1e6))
