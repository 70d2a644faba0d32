#!/usr/bin/env python3
"""Record-kernel time of one batch-1024 RSA-2048 call as a function of where its trace buffer starts inside one large
allocation (developer probe for the XCD-contiguous workgroup mapping: is its fast mode a matter of address alignment?).
usage: xcd_offset_probe.py [step MB] [count]   (H2R_LIB selects the build)"""
import os, sys, random
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
import halo2_rsa_amd as H
from halo2_rsa_amd import _lib
step = int(sys.argv[1]) if len(sys.argv) > 1 else 32
count = int(sys.argv[2]) if len(sys.argv) > 2 else 24
B = 1024
chip = H.BigIntChip(64, 2048)
pl = chip.pow_fixed_layout(65537)
rng = random.Random(5)
N = [rng.getrandbits(2048) | (1 << 2047) | 1 for _ in range(B)]
X = [rng.randrange(n) for n in N]
n, x = chip.assign_integer(N), chip.assign_integer(X)
need = B * pl.elem_stride
big = torch.zeros(need + step * (1 << 20) * count, dtype=torch.uint8, device="cuda")
print("base address 0x%x  (mod 2 MiB = %d KiB, mod 1 GiB = %d MiB)" % (big.data_ptr(), (big.data_ptr() % (2 << 20)) >> 10, (big.data_ptr() % (1 << 30)) >> 20))
ws = torch.zeros(chip.workspace_bytes(B, pl.num_mul_mods), dtype=torch.uint8, device="cuda")
for k in range(count):
    off = k * step * (1 << 20)
    tb = big[off:off + need]
    for _ in range(2):
        chip.pow_mod_fixed_exp(x, 65537, n, trace_buf=tb, workspace=ws)
    torch.cuda.synchronize()
    _lib.profile_enable(32)
    for _ in range(5):
        chip.pow_mod_fixed_exp(x, 65537, n, trace_buf=tb, workspace=ws)
    torch.cuda.synchronize()
    ms = _lib.profile_read(_lib.KERNEL_TRACE)
    _lib.profile_enable(0)
    avg = sum(ms) / len(ms)
    print("offset %5d MiB: record kernel %.4f ms  %.0f GB/s  (min %.4f max %.4f)" % (off >> 20, avg, B * 19 * 64338 / avg / 1e6, min(ms), max(ms)))
