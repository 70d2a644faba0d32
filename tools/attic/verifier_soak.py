#!/usr/bin/env python3
"""Soak of the pipelined RSASignatureVerifier (h2r_pipeline_signature_verifier): inside a step launch the chain role's encoded-message
check consumes the hashed limbs the SHA role of the SAME launch produces on other workgroups / XCDs (a count + acquire, no kernel
boundary in between).  Trains of six 1,024-signature calls over two buffer sets, a different message mix per call; after every train the
digests, limbs and verdicts of the last two calls are compared with hashlib and with the expected verdicts.  A stale or early read of
the limbs shows up as a wrong verdict.   usage: verifier_soak.py [trains]"""
import hashlib, os, random, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import json
import numpy as np
import torch
import halo2_rsa_amd as H
trains = int(sys.argv[1]) if len(sys.argv) > 1 else 40
kats = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden", "halo2_rsa_golden.json")))["rsa_kats"]
B = 1024
rsa = H.RSAChip(2048, 5)
chip = rsa.bigint_chip()
ns = [int(kats[i % 3]["n"]) for i in range(B)]
sigs = [int(kats[i % 3]["sig"]) for i in range(B)]
pk = rsa.assign_public_key(H.RSAPublicKey(H.UnassignedInteger.from_ints(ns, 32, 64), H.Fix(65537)))
sg = rsa.assign_signature(H.RSASignature(H.UnassignedInteger.from_ints(sigs, 32, 64)))
vl = rsa._verify_layout(pk)
pipe = H.Pipeline(chip, depth=2)
sets = [dict(trace=torch.zeros(B * vl.elem_stride, dtype=torch.uint8, device="cuda"), powed=torch.zeros((B, 32), dtype=torch.int64, device="cuda"),
             valid=torch.zeros(B, dtype=torch.uint8, device="cuda"), status=torch.zeros(B, dtype=torch.uint8, device="cuda"),
             hashed=torch.zeros((B, 4), dtype=torch.int64, device="cuda"), digest=torch.zeros((B, 32), dtype=torch.uint8, device="cuda"),
             ws=torch.zeros(chip.workspace_bytes(B, vl.pow.num_mul_mods), dtype=torch.uint8, device="cuda")) for _ in range(2)]
rng = random.Random(1)
bad = 0
for t in range(trains):
    calls = []
    for k in range(6):
        good = [rng.random() < 0.5 for _ in range(B)]
        msgs = [b"hello world" if g else bytes(rng.getrandbits(8) for _ in range(rng.choice([0, 3, 11, 55, 64, 128]))) for g in good]
        buf, off = H.pack_messages(msgs, torch.device("cuda", 0))
        b = sets[k & 1]
        pipe.signature_verifier(buf, off, 0, sg.c, 65537, pk.n, b["trace"], b["ws"], b["powed"], b["valid"], b["status"], b["hashed"], b["digest"])
        calls.append((msgs, buf, off))
    pipe.join()
    torch.cuda.synchronize()
    for k in (4, 5):
        b, msgs = sets[k & 1], calls[k][0]
        valid, digest = b["valid"].cpu().tolist(), b["digest"].cpu().numpy()
        for i in range(B):
            want = 1 if (msgs[i] == b"hello world" and i % 3 != 2) else 0
            if valid[i] != want or digest[i].tobytes() != hashlib.sha256(msgs[i]).digest():
                bad += 1
    if bad:
        break
print("verifier_soak: %d trains of 6 x %d signatures, %d mismatches" % (t + 1, B, bad))
sys.exit(1 if bad else 0)
