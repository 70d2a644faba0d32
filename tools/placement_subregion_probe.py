#!/usr/bin/env python3
"""Is the store-rate class of a big allocation inherited by its sub-regions?  K buffers of 12.4 GB (torch / hipMalloc) are classed by the
cells kernel (one 12.4 GB image each); then the RECORD kernel (trace_kernel, 1,024 RSA-2048 signatures = 1.25 GB, alone) is timed
into sub-regions carved at several offsets of every buffer.  If sub-regions follow their buffer's class, a trace arena can take its
regions from ONE large allocation chosen among a few, instead of looking at two dozen 1.25 GB candidates."""
import ctypes, os, random, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
import halo2_rsa_amd as H
from halo2_rsa_amd import _lib
L = _lib.lib()
B = 1024
chip = H.BigIntChip(64, 2048)
pl = chip.pow_fixed_layout(65537)
rng = random.Random(1)
N = [rng.getrandbits(2048) | (1 << 2047) | 1 for _ in range(B)]
X = [rng.randrange(n) for n in N]
n, x = chip.assign_integer(N), chip.assign_integer(X)
ws = torch.zeros(chip.workspace_bytes(B, pl.num_mul_mods), dtype=torch.uint8, device="cuda")
first = chip.pow_mod_fixed_exp(x, 65537, n, want_trace=False, workspace=ws)
rows = int(L.h2r_pow_advice_rows(chip._ctx, ctypes.byref(pl)))
img_bytes = B * rows * 160
region = B * pl.elem_stride
K = int(sys.argv[1]) if len(sys.argv) > 1 else 8
bufs = []
for i in range(K):
    bufs.append(torch.empty(img_bytes, dtype=torch.uint8, device="cuda"))
def cells_ms(buf):
    first.emit_advice(out=buf, direct=True)
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(2):
        first.emit_advice(out=buf, direct=True)
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / 2
def trace_ms(t):
    for _ in range(2):
        chip.pow_mod_fixed_exp(x, 65537, n, trace_buf=t, workspace=ws)
    torch.cuda.synchronize()
    _lib.profile_enable(16)
    for _ in range(4):
        chip.pow_mod_fixed_exp(x, 65537, n, trace_buf=t, workspace=ws)
    torch.cuda.synchronize()
    ms = _lib.profile_read(_lib.KERNEL_TRACE); _lib.profile_enable(0)
    return sum(ms) / len(ms)
offs = [0, 2, 4, 6, 8, 10]
print("buffer: cells kernel ms (TB/s) | record kernel alone, ms, into 1.25 GB sub-regions at offset GB %s" % offs)
for i, buf in enumerate(bufs):
    c = cells_ms(buf)
    line = []
    for o in offs:
        start = (o << 30)
        start += (-(buf.data_ptr() + start)) % 256
        if start + region > buf.numel():
            continue
        line.append("%.4f" % trace_ms(buf[start:start + region]))
    print("buffer %2d at %#x: %.3f (%.2f) | %s" % (i, buf.data_ptr(), c, img_bytes / c / 1e9, " ".join(line)), flush=True)
# plain 1.25 GB allocations for comparison
small = [torch.empty(region, dtype=torch.uint8, device="cuda") for _ in range(8)]
print("1.25 GB allocations: " + " ".join("%.4f" % trace_ms(t) for t in small))
