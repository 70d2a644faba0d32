#!/usr/bin/env python3
"""Record kernel (1,024 RSA-2048 signatures, alone) into the first 1.25 GB of allocations of different sizes, and into the same region
size carved at the END of them: at which allocation size does a sub-region stop behaving like an allocation of its own?"""
import os, random, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
import halo2_rsa_amd as H
from halo2_rsa_amd import _lib
B = 1024
chip = H.BigIntChip(64, 2048)
pl = chip.pow_fixed_layout(65537)
rng = random.Random(1)
N = [rng.getrandbits(2048) | (1 << 2047) | 1 for _ in range(B)]
X = [rng.randrange(n) for n in N]
n, x = chip.assign_integer(N), chip.assign_integer(X)
ws = torch.zeros(chip.workspace_bytes(B, pl.num_mul_mods), dtype=torch.uint8, device="cuda")
region = B * pl.elem_stride
def trace_ms(t):
    for _ in range(2):
        chip.pow_mod_fixed_exp(x, 65537, n, trace_buf=t, workspace=ws)
    torch.cuda.synchronize()
    _lib.profile_enable(16)
    for _ in range(4):
        chip.pow_mod_fixed_exp(x, 65537, n, trace_buf=t, workspace=ws)
    torch.cuda.synchronize()
    ms = _lib.profile_read(_lib.KERNEL_TRACE); _lib.profile_enable(0)
    return sum(ms) / len(ms)
keep = []
for mult in (1.0, 1.25, 1.5, 2, 3, 4, 8, 1.0):
    size = int(region * mult) + 4096
    row = []
    for i in range(4):
        b = torch.empty(size, dtype=torch.uint8, device="cuda")
        keep.append(b)
        a0 = (-b.data_ptr()) % 256
        e0 = size - region - 256
        e0 -= (b.data_ptr() + e0) % 256
        row.append("%.4f/%.4f" % (trace_ms(b[a0:a0 + region]), trace_ms(b[e0:e0 + region])))
    print("allocation = %.2f x region (%.2f GB): first / last region ms: %s" % (mult, size / 2**30, "  ".join(row)), flush=True)
