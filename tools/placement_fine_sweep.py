#!/usr/bin/env python3
"""Record kernel into a region whose START is swept in small steps inside one large allocation (2 MB-aligned base): does the offset of
the region inside a 2 MB page (and inside larger power-of-two blocks) decide the class?"""
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
buf = torch.empty(8 << 30, dtype=torch.uint8, device="cuda")
base = (-buf.data_ptr()) % (2 << 20)
print("allocation at %#x" % buf.data_ptr())
for name, offs in (("step 128 KB", [k * (128 << 10) for k in range(0, 40)]),
                   ("step 256 B", [k * 256 for k in range(0, 24)]),
                   ("step 4 KB", [k * 4096 for k in range(0, 40)]),
                   ("1 GB + step 192 KB", [(1 << 30) + k * (192 << 10) for k in range(0, 24)]),
                   ("odd multiples of 1 MB", [(2 * k + 1) << 20 for k in range(0, 24)]),
                   ("size - region - 256 of the earlier probe and neighbours", [int(region * 8) + 4096 - region - 256 - d for d in (0, 256, 4096, 65536, 1 << 20, 2 << 20)])):
    out = []
    for o in offs:
        o2 = base + o
        o2 -= o2 % 256
        if o2 + region > buf.numel():
            continue
        out.append("%d:%.3f" % (o, trace_ms(buf[o2:o2 + region])))
    print(name + " | " + " ".join(out), flush=True)
