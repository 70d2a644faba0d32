#!/usr/bin/env python3
"""BASELINE config 5's variable-exponent alternative: RSAPubE::Var with a 2,048-bit exponent per signature (32 limbs of 64 bits,
exp_limb_bits = 64) -- BigIntChip::pow_mod (big_integer/chip.rs:664-696) runs TWO mul_mods per exponent bit (multiply always,
select, square): 4,096 dependent mul_mods and 263 MB of witness per signature.  Batch 256 on one GPU, stream-ordered calls
(the variable-exponent path is not pipelined): ms per call, assigns/s, chain and record kernel times.
usage: var_exponent_timing.py [batch] [calls]"""
import os, sys, random, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import torch
import halo2_rsa_amd as H
from halo2_rsa_amd import _lib
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
CALLS = int(sys.argv[2]) if len(sys.argv) > 2 else 3
chip = H.BigIntChip(64, 2048)
rng = random.Random(0x68327273 + 5)
N = [rng.getrandbits(2048) | (1 << 2047) | 1 for _ in range(B)]
X = [rng.randrange(n) for n in N]
E = [rng.getrandbits(2048) | (1 << 2047) for _ in range(B)]
x, n, e = chip.assign_integer(X), chip.assign_integer(N), chip.assign_integer(E)
res = chip.pow_mod(x, e, n, 64)          # warm-up (code objects, allocator)
torch.cuda.synchronize()
got = res.value.to_big_uint()
assert all(got[i] == pow(X[i], E[i], N[i]) for i in (0, 1, B // 2, B - 1)) and not res.status.cpu().numpy().any()
bad, first = res.audit()                 # every record of every element, in place
torch.cuda.synchronize()
assert not bad.cpu().numpy().any()
sb = res.trace.stream_bytes
del res
_lib.profile_enable(512)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(CALLS):
    res = chip.pow_mod(x, e, n, 64)
    del res
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / CALLS
chain, trace = _lib.profile_read(_lib.KERNEL_CHAIN), _lib.profile_read(_lib.KERNEL_TRACE)
_lib.profile_enable(0)
print("pow_mod (Var, 2,048-bit exponents, 4,096 mul_mods per signature) batch %d: %.2f ms per call = %.0f assigns/s = %.2f TB/s of witness "
      "(%.1f MB per signature); chain kernel %.2f ms, record kernel %.2f ms (%.2f TB/s); every record audited in place: ok"
      % (B, dt * 1e3, B / dt, B * sb / dt / 1e12, sb / 1e6, sum(chain) / CALLS, sum(trace) / CALLS,
         B * 4096 * chip.layout.stream_bytes / (sum(trace) / CALLS) / 1e9)
      + " [per call: %d chain + %d record launches -- the exponent is walked as segments of its bits]" % (len(chain) // CALLS, len(trace) // CALLS))

# ---- the same through h2r_pipeline_modpow_public_key_var: call k+1's chains next to call k's record kernel (two buffer sets) ----
pl = chip.pow_var_layout(32, 64)
ies = chip.in_field_layout()[0]
mk = lambda nbytes: torch.empty(nbytes, dtype=torch.uint8, device="cuda")
sets = [dict(trace=mk(B * pl.elem_stride), inf=mk(B * ies), ws=mk(chip.workspace_bytes(B, pl.num_mul_mods)),
             out=torch.zeros((B, 32), dtype=torch.int64, device="cuda"), status=torch.zeros(B, dtype=torch.uint8, device="cuda")) for _ in range(2)]
pipe = chip.pipeline()
def call(k):
    s = sets[k % 2]
    pipe.modpow_public_key_var(x, e, 64, n, s["trace"], s["ws"], s["out"], s["status"], in_field_buf=s["inf"])
call(0); call(1); pipe.join(); torch.cuda.synchronize()
_lib.profile_enable(512)
K = 2 * CALLS
t0 = time.perf_counter()
for k in range(K):
    call(k)
pipe.join()
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / K
chain, trace = _lib.profile_read(_lib.KERNEL_CHAIN), _lib.profile_read(_lib.KERNEL_TRACE)
_lib.profile_enable(0)
got = H.AssignedInteger(sets[(K - 1) % 2]["out"], 64).to_big_uint()
assert all(got[i] == pow(X[i], E[i], N[i]) for i in (0, B - 1))
print("pipelined (h2r_pipeline_modpow_public_key_var, %d calls): %.2f ms per call = %.0f assigns/s = %.2f TB/s of witness; chain kernel %.2f ms, "
      "record kernel %.2f ms (%.2f TB/s)" % (K, dt * 1e3, B / dt, B * sb / dt / 1e12, sum(chain) / K, sum(trace) / K,
                                             B * 4096 * chip.layout.stream_bytes / (sum(trace) / K) / 1e9))
pipe.close()
