#!/usr/bin/env python3
"""Record kernel alone (1,024 RSA-2048 signatures) into (a) the first region of 9.5 GB allocations (physically contiguous extents: the slow
class) and (b) fresh allocations of 1.25 x the region, for the library given by H2R_LIB (record stride = 65,024 + 256 * pad bytes)."""
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
big = [torch.empty(int(8 * 1.25e9), dtype=torch.uint8, device="cuda") for _ in range(3)]
small = [torch.empty(int(region * 1.25), dtype=torch.uint8, device="cuda") for _ in range(4)]
gb = pl.num_mul_mods * B * chip.layout.stream_bytes / 1e9
fmt = lambda ts: " ".join("%.4f" % t for t in ts)
tb = [trace_ms(b[(-b.data_ptr()) % 256:][:region]) for b in big]
tl = [trace_ms(b[b.numel() - region - 4096 - ((b.data_ptr() + b.numel() - region - 4096) % 256):][:region]) for b in big]
ts = [trace_ms(b[(-b.data_ptr()) % 256:][:region]) for b in small]
print("record stride %d: big allocations, first region: %s | their last region: %s | small allocations: %s   (best %.2f TB/s, worst %.2f)" %
      (chip.layout.record_stride, fmt(tb), fmt(tl), fmt(ts), gb / min(tb + tl + ts) , gb / max(tb + tl + ts)))
