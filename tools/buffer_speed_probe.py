#!/usr/bin/env python3
"""Record-kernel time per trace BUFFER: N buffer sets allocated back to back, pipelined batch-1024 RSA-2048 calls rotating
through them, mean kernel time per buffer (the speed of the record kernel depends on where its output buffer lies).
usage: buffer_speed_probe.py [buffers] [calls]    (H2R_LIB / H2R_TRACE_MAP select build and mapping)"""
import os, sys, random
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
import halo2_rsa_amd as H
from halo2_rsa_amd import _lib
nset = int(sys.argv[1]) if len(sys.argv) > 1 else 6
CALLS = int(sys.argv[2]) if len(sys.argv) > 2 else 120
chip = H.BigIntChip(64, 2048); pl = chip.pow_fixed_layout(65537)
rng = random.Random(1); B = 1024
N = [rng.getrandbits(2048) | (1 << 2047) | 1 for _ in range(B)]; X = [rng.randrange(n) for n in N]
n, x = chip.assign_integer(N), chip.assign_integer(X)
import ctypes
ALLOC = os.environ.get("PROBE_ALLOC", "torch")   # torch | hipmalloc | contiguous | uncached (raw HIP allocations wrapped as tensors)
_hip = ctypes.CDLL("libamdhip64.so")
class _Raw:
    def __init__(self, ptr, n):
        self.__cuda_array_interface__ = {"shape": (n,), "typestr": "|u1", "data": (ptr, False), "version": 2}
def alloc_trace(nbytes):
    if ALLOC == "torch":
        return torch.zeros(nbytes, dtype=torch.uint8, device="cuda")
    if ALLOC.startswith("striped"):   # striped:<chunk MB>:<spread>
        _, ch, sp = ALLOC.split(":")
        v = ctypes.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "libvmm_alloc.so"))
        v.striped_alloc.restype = ctypes.c_void_p
        ptr = v.striped_alloc(ctypes.c_size_t(nbytes), ctypes.c_size_t(int(ch) << 20), ctypes.c_int(int(sp)), ctypes.c_int(0))
        assert ptr, "striped_alloc failed"
        t = torch.as_tensor(_Raw(ptr, nbytes), device="cuda")
        t.zero_()
        return t
    p = ctypes.c_void_p()
    if ALLOC == "hipmalloc":
        rc = _hip.hipMalloc(ctypes.byref(p), ctypes.c_size_t(nbytes))
    else:
        rc = _hip.hipExtMallocWithFlags(ctypes.byref(p), ctypes.c_size_t(nbytes), ctypes.c_uint(4 if ALLOC == "contiguous" else 3))
    assert rc == 0, (ALLOC, rc)
    t = torch.as_tensor(_Raw(p.value, nbytes), device="cuda")
    t.zero_()
    return t
print("allocation:", ALLOC)
if ALLOC == "arena":   # the library's own placement search: N buffers kept of 3N candidates
    ARENA = H.TraceArena.for_pow(chip, 65537, B, regions=nset, candidates=3 * nset)
    print("arena candidates (record kernel alone, ms):", " ".join("%.4f" % t for t in ARENA.measurements_ms))
    print("arena kept:", " ".join("%.4f" % t for t in ARENA.region_ms))
    _arena_it = iter(ARENA.regions)
    def alloc_trace(nbytes):
        return next(_arena_it)
sets = [dict(trace=alloc_trace(B * pl.elem_stride), ws=torch.zeros(chip.workspace_bytes(B, pl.num_mul_mods), dtype=torch.uint8, device="cuda"),
             out=torch.zeros((B, 32), dtype=torch.int64, device="cuda"), status=torch.zeros(B, dtype=torch.uint8, device="cuda")) for _ in range(nset)]
pipe = H.Pipeline(chip, depth=2, side_streams=1)
for k in range(2 * nset):
    s = sets[k % nset]; pipe.modpow_public_key(x, 65537, n, s["trace"], s["ws"], s["out"], s["status"])
pipe.join(); torch.cuda.synchronize()
_lib.profile_enable(4 * CALLS)
for k in range(CALLS):
    s = sets[k % nset]; pipe.modpow_public_key(x, 65537, n, s["trace"], s["ws"], s["out"], s["status"])
pipe.join(); torch.cuda.synchronize()
ms = _lib.profile_read(_lib.KERNEL_TRACE); _lib.profile_enable(0)
print("mapping", os.environ.get("H2R_TRACE_MAP", "default"))
for j in range(nset):
    v = ms[j::nset]
    print("buffer %d at 0x%x: record kernel mean %.4f ms (min %.4f max %.4f) %.0f GB/s" % (j, sets[j]["trace"].data_ptr(), sum(v) / len(v), min(v), max(v), B * 19 * 64338 / (sum(v) / len(v)) / 1e6))
