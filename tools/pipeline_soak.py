"""Soak run of the pipeline: 240 pipelined calls per configuration (1,024 / 3,072 signatures, 2 or 3 buffer sets, 1 or 2 record
streams), the oldest complete call audited in place every seventh call on the caller's stream.  usage: tools/pipeline_soak.py"""
import os, sys, random, torch, numpy as np
R = os.environ.get("GRAFT_REPO_ROOT", "/root/repo"); sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import halo2_rsa_amd as H
from halo2_rsa_amd import big_integer as BI
rng = random.Random(99)
# (bits, B, depth, side): RSA-2048 / RSA-1024 above 512 per call are one-launch steps (8,192: two launches per call), 384 per call the two-queue form;
# RSA-2048 calls of up to 2,048 on a pipeline with two side streams and three or more buffer sets: record kernels alternating between the side streams
# (last two: a 640-bit exponent on a small batch -- every call is walked as 5 segments of the exponent's bits, dense (two chains side by side) / sparse)
E_DENSE = rng.getrandbits(640) | (1 << 639)
E_SPARSE = (1 << 639) | (1 << 401) | (1 << 77) | 1
for bits, B, depth, side, E in ((2048, 1024, 2, 1, 65537), (2048, 3072, 2, 1, 65537), (2048, 1024, 3, 2, 65537), (2048, 2048, 3, 2, 65537), (2048, 640, 4, 2, 65537), (2048, 8192, 2, 1, 65537), (2048, 384, 2, 1, 65537),
                                (1024, 2048, 2, 1, 65537), (2048, 24, 2, 1, E_DENSE), (2048, 40, 3, 2, E_SPARSE),
                                # [r6] RSA-1024 in the two-queue form (one-wave chain kernels; 1,280: chain waves at issue priority; 8,192: sub-batches of 2,048)
                                (1024, 1280, 3, 2, 65537), (1024, 2048, 3, 2, 65537), (1024, 4096, 4, 2, 65537), (1024, 8192, 3, 2, 65537)):
    chip = H.BigIntChip(64, bits)
    pl = chip.pow_fixed_layout(E)
    base = [rng.getrandbits(bits) | (1 << (bits - 1)) | 1 for _ in range(64)]
    pipe = H.Pipeline(chip, depth=depth, side_streams=side)
    sets = [dict(trace=torch.zeros(B * pl.elem_stride, dtype=torch.uint8, device="cuda"),
                 ws=torch.zeros(chip.workspace_bytes(B, pl.num_mul_mods), dtype=torch.uint8, device="cuda"),
                 out=torch.zeros((B, bits // 64), dtype=torch.int64, device="cuda"),
                 status=torch.zeros(B, dtype=torch.uint8, device="cuda")) for _ in range(depth)]
    variants = []
    for v in range(4):
        N = [base[(i + v) % 64] ^ ((i // 64 + v) << (bits // 2 - 64)) | 1 for i in range(B)]
        X = [((base[(i * 5 + v) % 64] >> 3) * (i + 7 + v)) % N[i] for i in range(B)]
        variants.append((N, X, chip.assign_integer(N), chip.assign_integer(X)))
    bad_total, checks = 0, 0
    CALLS = 240 if B <= 3072 else 60
    e_bytes = E.to_bytes((E.bit_length() + 7) // 8, "little")
    for k in range(CALLS):
        v = variants[k % 4]; s = sets[k % depth]
        pipe.modpow_public_key(v[3], E, v[2], s["trace"], s["ws"], s["out"], s["status"])
        if k >= depth - 1 and k % 7 == 3:     # audit the oldest complete call on the caller's stream (ordered by the contract)
            kk = k - depth + 1
            vv = variants[kk % 4]; ss = sets[kk % depth]
            res = BI.BatchResult(H.AssignedInteger(ss["out"], 64), H.Trace(chip, ss["trace"], B, pl), ss["status"], workspace=ss["ws"],
                                 inputs=("pow_fixed", vv[3], None, vv[2], e_bytes))
            bad, first = res.audit()
            bad_total += int(bad.sum().item()) + int(ss["status"].sum().item()); checks += 1
            got = H.AssignedInteger(ss["out"].clone(), 64).to_big_uint()
            assert all(got[i] == pow(vv[1][i], E, vv[0][i]) for i in range(0, B, 37)), (B, k)
    pipe.join(); torch.cuda.synchronize()
    print("soak RSA-%d B=%d depth=%d streams=%d e=%d bits: %d calls, %d audits, violations %d" % (bits, B, depth, side, E.bit_length(), CALLS, checks, bad_total))
    assert bad_total == 0
    pipe.close()

# the advice image, pipelined (h2r_pipeline_modpow_public_key_advice): 90 calls over two image sets and two side streams, four input
# variants; every call's image is compared with the plain call's image of its variant right before its buffers are reused
import ctypes
chip = H.BigIntChip(64, 2048)
pl = chip.pow_fixed_layout(65537)
B, depth = 256, 2
sec = (ctypes.c_uint64 * 2)()
rows = int(H.lib().h2r_modpow_public_key_advice_rows(chip._ctx, ctypes.byref(pl), sec))
ifs = chip.in_field_layout()[0]
base = [rng.getrandbits(2048) | (1 << 2047) | 1 for _ in range(64)]
variants = []
for v in range(4):
    N = [base[(i + v) % 64] ^ ((i // 64 + v) << 900) | 1 for i in range(B)]
    X = [((base[(i * 5 + v) % 64] >> 3) * (i + 7 + v)) % N[i] for i in range(B)]
    x, n = chip.assign_integer(X), chip.assign_integer(N)
    plain = chip.pow_mod_fixed_exp(x, 65537, n, want_trace=False, check_in_field=True,
                                   workspace=torch.empty(chip.workspace_bytes(B, pl.num_mul_mods), dtype=torch.uint8, device="cuda"))
    variants.append((x, n, plain.emit_modpow_advice()))
sets = [dict(ws=torch.empty(chip.workspace_bytes(B, pl.num_mul_mods), dtype=torch.uint8, device="cuda"), out=torch.zeros((B, 32), dtype=torch.int64, device="cuda"),
             st=torch.zeros(B, dtype=torch.uint8, device="cuda"), inf=torch.zeros(B * ifs, dtype=torch.uint8, device="cuda"),
             img=torch.empty((B, rows * 160), dtype=torch.uint8, device="cuda")) for _ in range(depth)]
pipe = H.Pipeline(chip, depth, 2)
wrong = 0
CALLS = 90
for k in range(CALLS):
    s_ = sets[k % depth]
    if k >= depth:
        wrong += 0 if torch.equal(s_["img"], variants[(k - depth) % 4][2]) else 1
        s_["img"].fill_(0)
    pipe.modpow_public_key_advice(variants[k % 4][0], 65537, variants[k % 4][1], s_["ws"], s_["out"], s_["st"], s_["inf"], s_["img"])
pipe.join()
for k in range(CALLS - depth, CALLS):
    wrong += 0 if torch.equal(sets[k % depth]["img"], variants[k % 4][2]) else 1
torch.cuda.synchronize()
print("soak advice pipeline RSA-2048 B=%d depth=%d: %d calls, %d images compared, %d wrong" % (B, depth, CALLS, CALLS, wrong))
assert wrong == 0
pipe.close()
