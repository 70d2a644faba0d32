#!/usr/bin/env python3
"""Is the record kernel's speed in a region a property of the region's physical CHUNKS?  A pool of 256 MB chunks (HIP
virtual-memory API); regions of 5 chunks are mapped in several compositions and the record kernel (alone, batch 1024
RSA-2048) is timed in each: consecutive chunks, the same shifted by two, hybrids of the fastest and the slowest region,
and a permutation of one region's own chunks.  needs tools/libvmm_alloc.so (hipcc -shared tools/vmm_alloc.hip)."""
import os, sys, ctypes, random
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
import halo2_rsa_amd as H
from halo2_rsa_amd import _lib
V = ctypes.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "libvmm_alloc.so"))
V.pool_create.restype = ctypes.c_size_t; V.pool_chunk_bytes.restype = ctypes.c_size_t; V.pool_map.restype = ctypes.c_void_p
V.pool_unmap.argtypes = [ctypes.c_void_p, ctypes.c_int]
class _Raw:
    def __init__(self, ptr, n):
        self.__cuda_array_interface__ = {"shape": (n,), "typestr": "|u1", "data": (ptr, False), "version": 2}
B = 1024
chip = H.BigIntChip(64, 2048); pl = chip.pow_fixed_layout(65537)
rng = random.Random(1)
N = [rng.getrandbits(2048) | (1 << 2047) | 1 for _ in range(B)]; X = [rng.randrange(n) for n in N]
n, x = chip.assign_integer(N), chip.assign_integer(X)
ws = torch.zeros(chip.workspace_bytes(B, pl.num_mul_mods), dtype=torch.uint8, device="cuda")
need = B * pl.elem_stride
NCH = 48
got = V.pool_create(ctypes.c_size_t(NCH), ctypes.c_size_t(256 << 20), 0)
CH = V.pool_chunk_bytes(); per = (need + CH - 1) // CH
print("pool: %d chunks of %d MB, %d chunks per region" % (got, CH >> 20, per))
def measure(chunks):
    arr = (ctypes.c_int * len(chunks))(*chunks)
    base = V.pool_map(arr, len(chunks))
    assert base
    t = torch.as_tensor(_Raw(base, need), device="cuda")
    for _ in range(2):
        chip.pow_mod_fixed_exp(x, 65537, n, trace_buf=t, workspace=ws)
    torch.cuda.synchronize()
    _lib.profile_enable(16)
    for _ in range(4):
        chip.pow_mod_fixed_exp(x, 65537, n, trace_buf=t, workspace=ws)
    torch.cuda.synchronize()
    ms = _lib.profile_read(_lib.KERNEL_TRACE); _lib.profile_enable(0)
    del t
    torch.cuda.synchronize()
    assert V.pool_unmap(ctypes.c_void_p(base), len(chunks)) == 0
    return sum(ms) / len(ms)
regions = [list(range(per * i, per * i + per)) for i in range(got // per)]
times = [measure(r) for r in regions]
print("consecutive:", " ".join("%.4f" % t for t in times))
print("repeat     :", " ".join("%.4f" % measure(r) for r in regions))
sh = [[(c + 2) % got for c in r] for r in regions]
print("shifted by2:", " ".join("%.4f" % measure(r) for r in sh))
f = min(range(len(times)), key=lambda i: times[i]); s_ = max(range(len(times)), key=lambda i: times[i])
F, S = regions[f], regions[s_]
print("fastest region %d (%.4f)  slowest region %d (%.4f)" % (f, times[f], s_, times[s_]))
for k in range(per + 1):
    print("hybrid: first %d chunks of the fastest + last %d of the slowest: %.4f   |   the other way round: %.4f" % (k, per - k, measure(F[:k] + S[k:]), measure(S[:k] + F[k:])))
print("fastest, chunks reversed: %.4f   rotated by one: %.4f" % (measure(F[::-1]), measure(F[1:] + F[:1])))
print("slowest, chunks reversed: %.4f   rotated by one: %.4f" % (measure(S[::-1]), measure(S[1:] + S[:1])))
# every chunk's own contribution: region of the fastest with ONE chunk swapped for chunk c of the slowest, position by position
for pos in range(per):
    print("fastest with chunk %d replaced by the slowest's chunk %d: %.4f" % (pos, pos, measure(F[:pos] + [S[pos]] + F[pos + 1:])))
