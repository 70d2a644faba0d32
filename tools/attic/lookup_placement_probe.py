#!/usr/bin/env python3
"""lookup_fill_kernel's rate by OUTPUT BUFFER: the same 256-circuit call (10.7 GB of A' / S') written into several allocations
(all held, so that they are different memory) -- is the 5.1 TB/s the kernel or where its output lies?"""
import os, sys, random
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
import halo2_rsa_amd as H
from halo2_rsa_amd import _lib
B = 256
chip = H.BigIntChip(64, 2048)
rng = random.Random(5)
N = [rng.getrandbits(2048) | (1 << 2047) | 1 for _ in range(B)]
X = [rng.randrange(n) for n in N]
res = chip.pow_mod_fixed_exp(chip.assign_integer(X), 65537, chip.assign_integer(N))
la = H.LookupArgument(chip)
usable = (1 << 17) - 6
hist = la.new_hist(B)
la.hist_records(res.trace, hist)
P = 21888242871839275222246405745257275088548364400416034343698204186575808495617
thetas = [rng.randrange(P) for _ in range(B)]
del res
torch.cuda.empty_cache()
pairs = []
for i in range(int(sys.argv[1]) if len(sys.argv) > 1 else 6):
    out = (torch.empty((B, 5, usable, 32), dtype=torch.uint8, device="cuda"), torch.empty((B, 5, usable, 32), dtype=torch.uint8, device="cuda"))
    la.permuted_columns(hist, thetas, usable, out=out); torch.cuda.synchronize()
    _lib.profile_enable(16)
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(3):
        la.permuted_columns(hist, thetas, usable, out=out)
    b.record(); torch.cuda.synchronize()
    fill = _lib.profile_read(_lib.KERNEL_LOOKUP)
    _lib.profile_enable(0)
    gb = 2 * B * 5 * usable * 32 / 1e9
    print("buffer pair %d: call %.3f ms = %.2f TB/s; fill kernel %.3f ms = %.2f TB/s" % (i, a.elapsed_time(b) / 3, gb / (a.elapsed_time(b) / 3), sum(fill) / len(fill), gb / (sum(fill) / len(fill))))
    pairs.append(out)
