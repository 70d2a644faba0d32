#!/usr/bin/env python3
"""Round 3 probe: does the placement of the OTHER buffers of a pipelined call -- workspace (the chain role writes 20 MB of operands per
call, the record role reads them), in-field witness (10 MB), results -- matter, as the placement of the trace does?  On boxes whose
fresh memory is all slow-class the pipelined step runs 4-8 % slower than on other boxes although the kept trace regions are as fast
ALONE.  Here the trace regions stay the arena's, and the other buffers come either from torch (as in bench.py) or from physically
contiguous memory (hipExtMallocWithFlags(hipDeviceMallocContiguous): always the slow class for a trace, profiles/r02_xcd_mapping.txt).
A/B/A/B, 60 pipelined calls of 1,024 RSA-2048 signatures each; ms per step and the step launch average."""
import ctypes, os, sys, random, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
import halo2_rsa_amd as H
from halo2_rsa_amd import _lib
hip = ctypes.CDLL("libamdhip64.so")
class _Raw:
    def __init__(self, ptr, n):
        self.__cuda_array_interface__ = {"shape": (n,), "typestr": "|u1", "data": (ptr, False), "version": 2}
def contiguous(nbytes):
    p = ctypes.c_void_p()
    rc = hip.hipExtMallocWithFlags(ctypes.byref(p), ctypes.c_size_t(nbytes), ctypes.c_uint(0x4))   # hipDeviceMallocContiguous
    assert rc == 0, rc
    t = torch.as_tensor(_Raw(p.value, nbytes), device="cuda")
    t.zero_()
    return t
B = 1024
chip = H.BigIntChip(64, 2048); pl = chip.pow_fixed_layout(65537); ies = chip.in_field_layout()[0]
rng = random.Random(3)
N = [rng.getrandbits(2048) | (1 << 2047) | 1 for _ in range(B)]; X = [rng.randrange(n) for n in N]
n, x = chip.assign_integer(N), chip.assign_integer(X)
arena = H.TraceArena.for_pow(chip, 65537, B, regions=2, candidates=24)
print("kept regions (ms alone):", [round(t, 4) for t in arena.region_ms])
def make(kind):
    alloc = (lambda nb: torch.zeros(nb, dtype=torch.uint8, device="cuda")) if kind == "torch" else contiguous
    return [dict(ws=alloc(chip.workspace_bytes(B, pl.num_mul_mods)), inf=alloc(B * ies), out=alloc(B * 256).view(torch.int64).view(B, 32), status=alloc(B)) for _ in range(2)]
sets = {"torch": make("torch"), "contiguous": make("contiguous")}
pipe = chip.pipeline()
def run(kind, calls=60):
    ss = sets[kind]
    for k in range(4):
        s = ss[k % 2]; pipe.modpow_public_key(x, 65537, n, arena.regions[k % 2], s["ws"], s["out"], s["status"], in_field_buf=s["inf"])
    pipe.join(); torch.cuda.synchronize()
    _lib.profile_enable(256)
    t0 = time.perf_counter()
    for k in range(calls):
        s = ss[k % 2]; pipe.modpow_public_key(x, 65537, n, arena.regions[k % 2], s["ws"], s["out"], s["status"], in_field_buf=s["inf"])
    pipe.join(); torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / calls
    st = _lib.profile_read(_lib.KERNEL_STEP); _lib.profile_enable(0)
    print("%-11s %.4f ms per step, step launch %.4f ms (%d launches)" % (kind, dt * 1e3, sum(st) / len(st), len(st)), flush=True)
for rep in range(3):
    run("torch"); run("contiguous")
