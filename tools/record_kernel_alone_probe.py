#!/usr/bin/env python3
"""The record kernel ALONE (plain stream-ordered export, no overlap: developer build with H2R_PLAIN_OVERLAP=0), per shape: ms per launch and the
fraction of the HBM peak its records amount to, for a few plain allocations; with the -DH2R_ABLATION build, H2R_ABLATE = 1 (no carry phase) /
2 (no product loop) show what the stores wait for.  argv: bits batch [allocations = 4]
  H2R_LIB=halo2_rsa_amd/lib/variants/devabl.so H2R_PLAIN_OVERLAP=0 [H2R_ABLATE=n] python tools/record_kernel_alone_probe.py 1024 2048"""
import os, random, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import halo2_rsa_amd as H
from halo2_rsa_amd import _lib
bits, B = int(sys.argv[1]), int(sys.argv[2])
n_alloc = int(sys.argv[3]) if len(sys.argv) > 3 else 4
chip = H.BigIntChip(64, bits)
pl = chip.pow_fixed_layout(65537)
rng = random.Random(5)
base = [rng.getrandbits(bits) | (1 << (bits - 1)) | 1 for _ in range(64)]
N = [base[i % 64] for i in range(B)]
X = [(base[(i * 5) % 64] >> 3) * (i + 7) % N[i] for i in range(B)]
n, x = chip.assign_integer(N), chip.assign_integer(X)
rec_bytes = B * pl.num_mul_mods * chip.layout.stream_bytes if hasattr(chip.layout, "stream_bytes") else None
bufs = [torch.empty(B * pl.elem_stride, dtype=torch.uint8, device="cuda") for _ in range(n_alloc)]
out = []
for t in bufs:
    for _ in range(3):
        chip.pow_mod_fixed_exp(x, 65537, n, trace_buf=t)
    torch.cuda.synchronize()
    _lib.profile_enable(64)
    for _ in range(8):
        chip.pow_mod_fixed_exp(x, 65537, n, trace_buf=t)
    torch.cuda.synchronize()
    tr, ch = _lib.profile_read(_lib.KERNEL_TRACE), _lib.profile_read(_lib.KERNEL_CHAIN)
    _lib.profile_enable(0)
    out.append((sum(tr) / len(tr), sum(ch) / max(1, len(ch)), len(tr) / 8.0))
alg = B * pl.num_mul_mods * int(_lib.lib().h2r_algorithmic_bytes_per_mul_mod(chip._ctx)) if hasattr(_lib.lib(), "h2r_algorithmic_bytes_per_mul_mod") else None
for i, (tr, ch, per) in enumerate(out):
    print("RSA-%d B=%d alloc %d: record kernel %.4f ms (%.1f launches per call), chain kernel %.4f ms, region %.3f GB -> %.2f TB/s of region bytes" % (
        bits, B, i, tr, per, ch, B * pl.elem_stride / 1e9, B * pl.elem_stride / per / tr / 1e9))
