#!/usr/bin/env python3
"""Time the whole verify_pkcs1v15_signature witness (in-field + modpow + EM check) against modpow alone, and
the Fresh-op family, with the C ABI's per-kernel event profiler."""
import os, sys, random, hashlib
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
import halo2_rsa_amd as H
from halo2_rsa_amd import _lib
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
rsa = H.RSAChip(2048, 5)
chip = rsa.bigint_chip()
rng = random.Random(9)
N = [rng.getrandbits(2048) | (1 << 2047) | 1 for _ in range(B)]
S = [rng.randrange(n) for n in N]
Hh = [rng.getrandbits(256) for _ in range(B)]
pk = H.RSAPublicKey(H.UnassignedInteger.from_ints(N, 32, 64), H.Fix(65537))
sg = H.RSASignature(H.UnassignedInteger.from_ints(S, 32, 64))
def timed(fn, reps=10):
    fn(); torch.cuda.synchronize()
    _lib.profile_enable(64)
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): fn()
    b.record(); torch.cuda.synchronize()
    ks = {n: _lib.profile_read(k) for n, k in (("chain", _lib.KERNEL_CHAIN), ("trace", _lib.KERNEL_TRACE), ("aux", _lib.KERNEL_AUX))}
    _lib.profile_enable(0)
    return a.elapsed_time(b) / reps, {n: (round(sum(v) / len(v), 4) if v else None) for n, v in ks.items()}
print("verify_pkcs1v15 (incl. host-side assign + allocations):", timed(lambda: rsa.verify_pkcs1v15_signature(pk, Hh, sg)))
an, asg = chip.assign_integer(pk.n), chip.assign_integer(sg.c)
print("pow_mod_fixed_exp (device-resident inputs):", timed(lambda: chip.pow_mod_fixed_exp(asg, 65537, an, check_in_field=True)))
for op in ("add", "sub", "add_mod", "sub_mod", "is_less_than", "is_in_field"):
    f = getattr(chip, op)
    args = (asg, an, an) if op in ("add_mod", "sub_mod") else (asg, an)
    print(op, timed(lambda: f(*args)))
