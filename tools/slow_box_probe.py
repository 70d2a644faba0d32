#!/usr/bin/env python3
"""On a box where the arena finds no fast region: does any other way of obtaining memory?  Record kernel alone (batch 1024
RSA-2048), best and median of 12 regions per strategy: the arena's own candidates, torch allocations, regions mapped after
a 100 GB dummy allocation (another part of the physical memory), pool regions of 8 MB burst-created chunks."""
import os, sys, ctypes, random
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
import halo2_rsa_amd as H
from halo2_rsa_amd import _lib
B = 1024
chip = H.BigIntChip(64, 2048); pl = chip.pow_fixed_layout(65537)
rng = random.Random(1)
N = [rng.getrandbits(2048) | (1 << 2047) | 1 for _ in range(B)]; X = [rng.randrange(n) for n in N]
n, x = chip.assign_integer(N), chip.assign_integer(X)
ws = torch.zeros(chip.workspace_bytes(B, pl.num_mul_mods), dtype=torch.uint8, device="cuda")
need = B * pl.elem_stride
def time_in(t):
    for _ in range(2):
        chip.pow_mod_fixed_exp(x, 65537, n, trace_buf=t, workspace=ws)
    torch.cuda.synchronize()
    _lib.profile_enable(16)
    for _ in range(3):
        chip.pow_mod_fixed_exp(x, 65537, n, trace_buf=t, workspace=ws)
    torch.cuda.synchronize()
    ms = _lib.profile_read(_lib.KERNEL_TRACE); _lib.profile_enable(0)
    return sum(ms) / len(ms)
def report(name, v):
    v = sorted(v)
    print("%-34s best %.4f  second %.4f  median %.4f  worst %.4f  (%d below 0.19)" % (name, v[0], v[1], v[len(v) // 2], v[-1], sum(1 for t in v if t < 0.19)))
a = H.TraceArena.for_pow(chip, 65537, B, regions=1, candidates=12)
report("arena candidates (alone, library)", a.measurements_ms); a.close()
bufs = [torch.zeros(need, dtype=torch.uint8, device="cuda") for _ in range(12)]
report("torch allocations", [time_in(b) for b in bufs]); del bufs; torch.cuda.empty_cache()
dummy = torch.empty(100 << 30, dtype=torch.uint8, device="cuda")
a = H.TraceArena.for_pow(chip, 65537, B, regions=1, candidates=12)
report("arena behind a 100 GB allocation", a.measurements_ms); a.close()
bufs = [torch.zeros(need, dtype=torch.uint8, device="cuda") for _ in range(12)]
report("torch behind a 100 GB allocation", [time_in(b) for b in bufs]); del bufs
dummy2 = torch.empty(100 << 30, dtype=torch.uint8, device="cuda")
a = H.TraceArena.for_pow(chip, 65537, B, regions=1, candidates=12)
report("arena behind 200 GB", a.measurements_ms); a.close()
del dummy, dummy2; torch.cuda.empty_cache()
V = ctypes.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "libvmm_alloc.so"))
V.pool_create.restype = ctypes.c_size_t; V.pool_chunk_bytes.restype = ctypes.c_size_t; V.pool_map.restype = ctypes.c_void_p
V.pool_unmap.argtypes = [ctypes.c_void_p, ctypes.c_int]
class _Raw:
    def __init__(self, ptr, nb):
        self.__cuda_array_interface__ = {"shape": (nb,), "typestr": "|u1", "data": (ptr, False), "version": 2}
got = V.pool_create(ctypes.c_size_t((16 << 30) // (8 << 20)), ctypes.c_size_t(8 << 20), 0)
CH = V.pool_chunk_bytes(); per = (need + CH - 1) // CH
out = []
for i in range(12):
    arr = (ctypes.c_int * per)(*range(per * i, per * i + per))
    base = V.pool_map(arr, per)
    t = torch.as_tensor(_Raw(base, need), device="cuda")
    out.append(time_in(t)); del t; torch.cuda.synchronize()
    V.pool_unmap(ctypes.c_void_p(base), per)
report("pool of 8 MB chunks (burst)", out)
