"""Debug: per-phase s_memtime stamps of chain_kernel block 0.
Needs the instrumented build:  python -m halo2_rsa_amd._build timing -DH2R_CHAIN_TIMING -DH2R_DEV_KNOBS  (run before gpurun)."""
import os, sys
os.environ["H2R_CHAIN_TIMING"] = "1"
os.environ.setdefault("H2R_LIB", os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "halo2_rsa_amd", "lib", "variants", "timing.so"))
sys.path.insert(0, ".")
import torch, random
import halo2_rsa_amd as H
chip = H.BigIntChip(64, 2048)
rng = random.Random(1)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
N = [rng.getrandbits(2048) | (1 << 2047) | 1 for _ in range(B)]
X = [rng.randrange(n) for n in N]
for _ in range(2):
    chip.pow_mod_fixed_exp(chip.assign_integer(X), 65537, chip.assign_integer(N), want_trace=False)
torch.cuda.synchronize()
t = [int(l) for l in open("/tmp/h2r_chain_timing.txt")]
d = [b - a for a, b in zip(t, t[1:])]
# 5 stamps per block_mul: [pre-B1, post-B1, pre-B2(after products), post-B2, end(reduce + carries by wave 0)]
# (each stamp itself costs ~265 cycles: subtract that from every interval)
names = ["B1 wait", "products", "B2 wait", "reduce+carries", "glue->next"]
NS = len(names)
n = len(t) // NS
print("stamps", len(t), "block_muls", n, "total cycles", t[-1] - t[0], "(s_memtime ticks ~ shader cycles)")
import collections
acc = collections.defaultdict(list)
for k in range(n):
    for j in range(NS):
        idx = NS * k + j
        if idx < len(d):
            acc[(k % 3, names[j])].append(d[idx])
for mode, mn in enumerate(["FULL", "HIGH", "LOW"]):
    print(mn, {nm: round(sum(acc[(mode, nm)]) / max(1, len(acc[(mode, nm)])), 1) for nm in names})
