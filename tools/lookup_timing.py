#!/usr/bin/env python3
"""Time the lookup range-check kernels (multiplicity histogram, grouped permutation) on a config-2 trace."""
import os, sys, random, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
import halo2_rsa_amd as H
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
chip = H.BigIntChip(64, 2048)
rng = random.Random(5)
N = [rng.getrandbits(2048) | (1 << 2047) | 1 for _ in range(B)]
X = [rng.randrange(n) for n in N]
res = chip.pow_mod_fixed_exp(chip.assign_integer(X), 65537, chip.assign_integer(N))
torch.cuda.synchronize()
tr = res.trace
lookups = 20330 * B
for name, fn in (("hist", tr.lookup_hist), ("perm", tr.lookup_permutation)):
    fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(10):
        fn()
    b.record(); torch.cuda.synchronize()
    ms = a.elapsed_time(b) / 10
    print("%s: %.3f ms per batch of %d  (%.1f G lookups/s; incl. output allocation)" % (name, ms, B, lookups / ms / 1e6))
