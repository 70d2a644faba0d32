#!/usr/bin/env python3
"""Record kernel (1,024 RSA-2048 signatures, alone) into a 1.19 GB region at a sweep of start offsets inside ONE large allocation: is the
store-rate class a function of where the region starts (physical address bits), and with which period?
usage: placement_offset_sweep.py [alloc_GB] [stride_MB] [fine_lo_MB fine_hi_MB fine_stride_MB]"""
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
    chip.pow_mod_fixed_exp(x, 65537, n, trace_buf=t, workspace=ws)
    torch.cuda.synchronize()
    _lib.profile_enable(8)
    for _ in range(3):
        chip.pow_mod_fixed_exp(x, 65537, n, trace_buf=t, workspace=ws)
    torch.cuda.synchronize()
    ms = _lib.profile_read(_lib.KERNEL_TRACE); _lib.profile_enable(0)
    return min(ms)
gb = float(sys.argv[1]) if len(sys.argv) > 1 else 16
stride = int(float(sys.argv[2]) * 2**20) if len(sys.argv) > 2 else 128 << 20
for rep in range(2):
    buf = torch.empty(int(gb * 2**30), dtype=torch.uint8, device="cuda")
    base = (-buf.data_ptr()) % (2 << 20)
    print("allocation %d at %#x (%.1f GB), region %.3f GB, stride %d MB" % (rep, buf.data_ptr(), gb, region / 2**30, stride >> 20))
    out = []
    o = base
    while o + region <= buf.numel():
        out.append((o, trace_ms(buf[o:o + region])))
        o += stride
    print(" ".join("%d:%.3f" % (oo >> 20, t) for oo, t in out), flush=True)
    if len(sys.argv) > 5 and rep == 0:
        lo, hi, st = (int(float(v) * 2**20) for v in sys.argv[3:6])
        o = base + lo
        fine = []
        while o <= base + hi and o + region <= buf.numel():
            fine.append((o, trace_ms(buf[o:o + region])))
            o += st
        print("fine: " + " ".join("%.2f:%.3f" % (oo / 2**20, t) for oo, t in fine), flush=True)
