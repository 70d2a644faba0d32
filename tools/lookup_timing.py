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

# ---- halo2's lookup argument (h2r_lookup_*): per-argument multiplicities, then the permuted columns A' / S' ----
from halo2_rsa_amd import _lib
la = H.LookupArgument(chip)
usable = (1 << 17) - 6                      # k = 17: one RSA-2048 modpow_public_key circuit (75.5 k rows) fits
CH = min(B, 256)                            # circuits per call: 256 x 5 arguments x 2 columns x 4.19 MB = 10.7 GB
hist = la.new_hist(B)
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
la.hist_records(tr, hist); torch.cuda.synchronize(); hist.zero_()
a.record(); la.hist_records(tr, hist); b.record(); torch.cuda.synchronize()
print("lookup_hist_records: %.3f ms per batch of %d (per-argument multiplicities, %d table rows)" % (a.elapsed_time(b), B, la.n_rows))
P = 21888242871839275222246405745257275088548364400416034343698204186575808495617
thetas = [rng.randrange(P) for _ in range(B)]
out = (torch.empty((CH, 5, usable, 32), dtype=torch.uint8, device="cuda"), torch.empty((CH, 5, usable, 32), dtype=torch.uint8, device="cuda"))
la.permuted_columns(hist[:CH].contiguous(), thetas[:CH], usable, out=out); torch.cuda.synchronize()
_lib.profile_enable(64)
t0 = time.perf_counter()
for c0 in range(0, B, CH):
    la.permuted_columns(hist[c0:c0 + CH].contiguous(), thetas[c0:c0 + CH], usable, out=out)
torch.cuda.synchronize()
wall = (time.perf_counter() - t0) * 1e3
fill = _lib.profile_read(_lib.KERNEL_LOOKUP)
_lib.profile_enable(0)
bytes_per_call = CH * 5 * 2 * usable * 32
print("lookup_permuted_columns: %d circuits x 5 arguments x (A', S') x %d rows: %.2f ms wall for the batch; lookup_fill_kernel %.3f ms per %d "
      "circuits = %.2f TB/s written (%.1f GB per call; setup kernel + fill kernel, wall %.3f ms per call)" %
      (B, usable, wall, sum(fill) / len(fill), CH, bytes_per_call / (sum(fill) / len(fill)) / 1e9, bytes_per_call / 1e9, wall / (B / CH)))
