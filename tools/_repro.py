import sys, time, random, torch, numpy as np
import os; R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, os.path.join(R, "tests")); sys.path.insert(0, R)
import halo2_rsa_amd as H
from test_gpu_parity import rand_modulus
mode = sys.argv[1]
if "userstream" in mode:
    torch.cuda.set_stream(torch.cuda.Stream())
chip = H.BigIntChip(64, 2048)
depth = 2
pipe = H.Pipeline(chip, depth=depth, side_streams=1)
pl = chip.pow_fixed_layout(65537)
rng = random.Random(4244)
B, CALLS = 1024, 5
sets = [dict(trace=torch.empty(B * pl.elem_stride, dtype=torch.uint8, device="cuda"),
             ws=torch.empty(chip.workspace_bytes(B, pl.num_mul_mods), dtype=torch.uint8, device="cuda"),
             out=torch.empty((B, 32), dtype=torch.int64, device="cuda"),
             status=torch.zeros(B, dtype=torch.uint8, device="cuda")) for _ in range(depth)]
base_n = [rand_modulus(rng, 2048) for _ in range(B)]
inputs, snaps = [], {}
for k in range(CALLS):
    N = base_n[k:] + base_n[:k]
    X = [(n >> (k + 1)) % n for n in N]
    inputs.append((N, X, chip.assign_integer(N), chip.assign_integer(X)))
if "prealloc" in mode:
    pre = [(torch.empty_like(sets[0]["trace"]), torch.empty_like(sets[0]["out"]), torch.empty_like(sets[0]["status"])) for _ in range(CALLS)]
torch.cuda.synchronize()
t0 = time.perf_counter()
for k in range(CALLS):
    s = sets[k % depth]
    if k >= depth and "noclone" not in mode:
        if "prealloc" in mode:
            for d, src in zip(pre[k - depth], (s["trace"], s["out"], s["status"])):
                if "addcopy" in mode: torch.add(src, 0, out=d)
                elif "small" in mode and src.numel() > 100000: pass
                else: d.copy_(src)
            snaps[k - depth] = pre[k - depth]
        else:
            snaps[k - depth] = (s["trace"].clone(), s["out"].clone(), s["status"].clone())
    pipe.modpow_public_key(inputs[k][3], 65537, inputs[k][2], s["trace"], s["ws"], s["out"], s["status"])
    print("issued", k, "%.3f ms" % (1e3 * (time.perf_counter() - t0)), flush=True)
pipe.join()
torch.cuda.synchronize()
print(mode, "total %.1f ms" % (1e3 * (time.perf_counter() - t0)))
for k, s in enumerate(sets):
    st = s["status"].cpu().numpy()
    print(" set", k, "status!=0:", int((st != 0).sum()))
for k, v in snaps.items():
    st = v[2].cpu().numpy()
    print(" snap", k, "status!=0:", int((st != 0).sum()))
