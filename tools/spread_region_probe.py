#!/usr/bin/env python3
"""Regions whose XCD eighths lie far apart physically: a pool of 160 MB chunks (8 areas x K chunks, created back to back);
"consecutive" region r = chunks 8r .. 8r+7, "spread" region r = chunk r of every area (the eight XCD streams then write
K x 160 MB apart).  Record kernel alone, batch 1024 RSA-2048.  needs tools/libvmm_alloc.so."""
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
K = int(sys.argv[1]) if len(sys.argv) > 1 else 8
got = V.pool_create(ctypes.c_size_t(8 * K), ctypes.c_size_t(160 << 20), 0)
CH = V.pool_chunk_bytes()
assert got == 8 * K and 8 * CH >= need
def measure(chunks):
    arr = (ctypes.c_int * len(chunks))(*chunks)
    base = V.pool_map(arr, len(chunks)); assert base
    t = torch.as_tensor(_Raw(base, need), device="cuda")
    for _ in range(2):
        chip.pow_mod_fixed_exp(x, 65537, n, trace_buf=t, workspace=ws)
    torch.cuda.synchronize()
    _lib.profile_enable(16)
    for _ in range(4):
        chip.pow_mod_fixed_exp(x, 65537, n, trace_buf=t, workspace=ws)
    torch.cuda.synchronize()
    ms = _lib.profile_read(_lib.KERNEL_TRACE); _lib.profile_enable(0)
    del t; torch.cuda.synchronize()
    assert V.pool_unmap(ctypes.c_void_p(base), len(chunks)) == 0
    return sum(ms) / len(ms)
print("pool of %d chunks of %d MB" % (got, CH >> 20))
print("consecutive:", " ".join("%.4f" % measure(list(range(8 * r, 8 * r + 8))) for r in range(K)))
print("spread     :", " ".join("%.4f" % measure([a * K + r for a in range(8)]) for r in range(K)))
print("spread, areas reversed:", " ".join("%.4f" % measure([(7 - a) * K + r for a in range(8)]) for r in range(K)))
print("stride 2   :", " ".join("%.4f" % measure([(2 * a + 16 * r) % (8 * K) + (1 if (2 * a + 16 * r) // (8 * K) % 2 else 0) for a in range(8)]) for r in range(K // 2)))
