#!/usr/bin/env python3
"""Record-kernel time (alone, batch 1024 RSA-2048) in regions mapped from a chunk pool with different chunk sizes and
virtual-address alignments: is the buffer dependence of the kernel's speed a matter of page-table fragment size
(min of the VA alignment and the physical block)?  needs tools/libvmm_alloc.so."""
import os, sys, ctypes, random
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
import halo2_rsa_amd as H
from halo2_rsa_amd import _lib
V = ctypes.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "libvmm_alloc.so"))
V.pool_create.restype = ctypes.c_size_t; V.pool_chunk_bytes.restype = ctypes.c_size_t; V.pool_map.restype = ctypes.c_void_p
V.pool_unmap.argtypes = [ctypes.c_void_p, ctypes.c_int]; V.pool_set_va_alignment.argtypes = [ctypes.c_size_t]
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
chunk_mb = int(sys.argv[1]) if len(sys.argv) > 1 else 256
got = V.pool_create(ctypes.c_size_t((int(os.environ.get('POOL_GB', '12')) << 30) // (chunk_mb << 20)), ctypes.c_size_t(chunk_mb << 20), 0)
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
    return sum(ms) / len(ms), base
regions = [list(range(per * i, per * i + per)) for i in range(min(8, got // per))]
for align_mb in ((0,) if os.environ.get("ONE_ALIGN") else (0, 2, 64, 256, 1024, 4096)):
    V.pool_set_va_alignment(ctypes.c_size_t(align_mb << 20))
    out = []
    for r in regions:
        t, base = measure(r)
        out.append("%.4f@%dM" % (t, (base % (1 << 32)) >> 20))
    print("VA alignment %5d MB:" % align_mb, " ".join(out))
