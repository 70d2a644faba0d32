#!/usr/bin/env python3
"""RSA-1024 chain kernel alone (no records): time per call against the number of dependent mul_mods (exponents 1, 3, 17, 257, 65537, 2^32+1) and the
batch -- the fixed cost per element (modulus set-up: the Knuth-D reciprocal) apart from the cost per mul_mod, for the one-wave chain (H2R_CHAIN_WAVE=1,
default) and the four-wave chain (=0; developer build).  python tools/chain_wave_cost_probe.py"""
import os, random, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
import halo2_rsa_amd as H
from halo2_rsa_amd import _lib
bits = 1024
chip = H.BigIntChip(64, bits)
rng = random.Random(3)
for B in (256, 1024, 2048, 4096):
    N = [rng.getrandbits(bits) | (1 << (bits - 1)) | 1 for _ in range(B)]
    X = [rng.randrange(n) for n in N]
    x, n = chip.assign_integer(X), chip.assign_integer(N)
    pts = []
    for e in (1, 3, 17, 257, 65537, (1 << 32) + 1):
        T = chip.pow_fixed_layout(e).num_mul_mods
        ws = torch.empty(chip.workspace_bytes(B, T), dtype=torch.uint8, device="cuda")
        for _ in range(3):
            chip.pow_mod_fixed_exp(x, e, n, want_trace=False, workspace=ws)
        torch.cuda.synchronize()
        _lib.profile_enable(64)
        for _ in range(8):
            chip.pow_mod_fixed_exp(x, e, n, want_trace=False, workspace=ws)
        torch.cuda.synchronize()
        ms = _lib.profile_read(_lib.KERNEL_CHAIN)
        _lib.profile_enable(0)
        pts.append((T, 1e3 * sum(ms) / 8))
    (t0, u0), (t1, u1) = pts[0], pts[-1]
    slope = (u1 - u0) / (t1 - t0)
    print("batch %5d  chain_wave=%s  " % (B, os.environ.get("H2R_CHAIN_WAVE", "default")) + "  ".join("T=%d: %.1f us" % p for p in pts) +
          "   => %.2f us per dependent mul_mod + %.1f us per element besides" % (slope, u0 - slope * t0))
