#!/usr/bin/env python3
"""Time sha256_kernel (h2r_sha256.hpp) -- the caller-side step of RSASignatureVerifier (reference src/lib.rs:205-239) -- and the
whole h2r_signature_verifier_batch next to h2r_verify_pkcs1v15_batch on precomputed digests.
usage: sha256_timing.py [msg_len]"""
import os, sys, random, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import torch
import halo2_rsa_amd as H
from halo2_rsa_amd import _lib
from halo2_rsa_amd._lib import lib
n = int(sys.argv[1]) if len(sys.argv) > 1 else 128
chip = H.BigIntChip(64, 2048)
for B in (1024, 8192, 65536, 1 << 20):
    raw = torch.randint(0, 256, (B * n,), dtype=torch.uint8, device="cuda")
    digest = torch.empty((B, 32), dtype=torch.uint8, device="cuda")
    hashed = torch.empty((B, 4), dtype=torch.int64, device="cuda")
    hm = torch.empty((B, 288), dtype=torch.uint8, device="cuda")
    st = chip._stream()
    call = lambda: lib().h2r_sha256_hashed_msg_batch(chip._ctx, raw.data_ptr(), None, n, B, digest.data_ptr(), hashed.data_ptr(), hm.data_ptr(), 288, st)
    for _ in range(3): assert call() == 0
    torch.cuda.synchronize()
    _lib.profile_enable(32)
    for _ in range(10): call()
    torch.cuda.synchronize()
    ms = _lib.profile_read(_lib.KERNEL_SHA256); _lib.profile_enable(0)
    avg = sum(ms) / len(ms)
    blocks = (n + 9 + 63) // 64
    print("sha256_kernel batch %7d x %d-byte messages (%d blocks each): %.4f ms per launch (min %.4f) = %.1f M messages/s, %.2f G compressions/s, %.1f GB/s of message bytes"
          % (B, n, blocks, avg, min(ms), B / avg / 1e3, B * blocks / avg / 1e6, B * n / avg / 1e6))
# whole verifier from message bytes vs the verifier on precomputed digests (batch 1,024, RSA-2048, e = 65537)
rsa = H.RSAChip(2048, 5)
rng = random.Random(3)
B = 1024
nmod = rng.getrandbits(2048) | (1 << 2047) | 1
pk = rsa.assign_public_key(H.RSAPublicKey(H.UnassignedInteger.from_ints([nmod], 32, 64), H.Fix(65537)))
sg = rsa.assign_signature(H.RSASignature(H.UnassignedInteger.from_ints([rng.randrange(nmod) for _ in range(B)], 32, 64)))
msgs = H.pack_messages([bytes(rng.getrandbits(8) for _ in range(n)) for _ in range(B)], torch.device("cuda", 0))
ver = H.RSASignatureVerifier(rsa)
def wall(fn, reps=5):
    fn(); torch.cuda.synchronize()
    best = 1e9
    for _ in range(reps):
        t0 = time.perf_counter(); r = fn(); torch.cuda.synchronize(); best = min(best, time.perf_counter() - t0); del r
    return best * 1e3
t_msg = wall(lambda: ver.verify_pkcs1v15_signature(pk, msgs, sg))
res = ver.verify_pkcs1v15_signature(pk, msgs, sg)
hashed_dev = H.AssignedInteger(res.inputs[2], 64)
t_dig = wall(lambda: rsa.verify_pkcs1v15_signature(pk, hashed_dev, sg))
print("RSASignatureVerifier batch %d from %d-byte messages: %.3f ms wall (min of 5, torch.empty of the buffers included); verify on precomputed digests: %.3f ms"
      % (B, n, t_msg, t_dig))
