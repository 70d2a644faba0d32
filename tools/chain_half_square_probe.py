#!/usr/bin/env python3
"""Chain kernel time per call (no records) for the shipped library and for the -DH2R_ABL_HALF_FULL_PRODUCT variant (the full product's
loop at half length: wrong results, timing only = the ceiling of a squaring-specific product).  python tools/chain_half_square_probe.py"""
import os, subprocess, sys
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
CODE = """
import sys, random, torch
sys.path.insert(0, %r)
import halo2_rsa_amd as H
from halo2_rsa_amd import _lib
bits, B, ebits = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
chip = H.BigIntChip(64, bits)
rng = random.Random(3)
N = [rng.getrandbits(bits) | (1 << (bits - 1)) | 1 for _ in range(B)]
X = [rng.randrange(n) for n in N]
e = 65537 if ebits == 17 else (rng.getrandbits(ebits) | (1 << (ebits - 1)))
x, n = chip.assign_integer(X), chip.assign_integer(N)
T = chip.pow_fixed_layout(e).num_mul_mods
ws = torch.empty(chip.workspace_bytes(B, T), dtype=torch.uint8, device='cuda')
for _ in range(3): chip.pow_mod_fixed_exp(x, e, n, want_trace=False, workspace=ws)
torch.cuda.synchronize()
_lib.profile_enable(64)
for _ in range(8): chip.pow_mod_fixed_exp(x, e, n, want_trace=False, workspace=ws)
torch.cuda.synchronize()
ms = _lib.profile_read(_lib.KERNEL_CHAIN)
print('%%d-bit batch %%d, %%d mul_mods: chain kernels %%.4f ms per call (%%d launches per call) -> %%.3f us per dependent mul_mod' %% (bits, B, T, sum(ms) / 8, len(ms) // 8, 1e3 * sum(ms) / 8 / T))
""" % ROOT
for lib in ("shipped", "halfsq"):
    env = dict(os.environ)
    if lib != "shipped":
        env["H2R_LIB"] = os.path.join(ROOT, "halo2_rsa_amd", "lib", "variants", lib + ".so")
    for bits, B, eb in ((1024, 1024, 17), (2048, 1024, 17), (2048, 256, 2048)):
        out = subprocess.run([sys.executable, "-c", CODE, str(bits), str(B), str(eb)], capture_output=True, text=True, timeout=300, env=env)
        print(lib.ljust(8), (out.stdout.strip().splitlines() or [out.stderr[-300:]])[-1], flush=True)
