"""Which allocations are of which placement class?  16 buffers of the lookup column size; class of buffer i = whether h2r_lookup_permuted_columns with
(A' = pool[0], S' = pool[i]) is fast (the other class than pool[0]) or slow (pool[0]'s class); printed next to the buffers' virtual addresses."""
import os, random, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import halo2_rsa_amd as H

B = 256
chip = H.BigIntChip(64, 2048)
rng = random.Random(1)
N = [rng.getrandbits(2048) | (1 << 2047) | 1 for _ in range(B)]
X = [rng.randrange(n) for n in N]
res = chip.pow_mod_fixed_exp(chip.assign_integer(X), 65537, chip.assign_integer(N))
la = H.LookupArgument(chip)
usable = (1 << 17) - 6
hist = la.new_hist(B)
la.hist_records(res.trace, hist)
torch.cuda.synchronize()
del res
torch.cuda.empty_cache()
P = 21888242871839275222246405745257275088548364400416034343698204186575808495617
thetas = [rng.randrange(P) for _ in range(B)]
n = int(sys.argv[1]) if len(sys.argv) > 1 else 16
pool = [torch.empty((B, 5, usable, 32), dtype=torch.uint8, device="cuda") for _ in range(n)]


def t(ia, is_):
    la.permuted_columns(hist, thetas, usable, out=(pool[ia], pool[is_]))
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(2):
        la.permuted_columns(hist, thetas, usable, out=(pool[ia], pool[is_]))
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / 2
ms = [None] + [t(0, i) for i in range(1, n)]
thr = (min(ms[1:]) + max(ms[1:])) / 2
cls = ["X"] + [("Y" if m < thr else "X") for m in ms[1:]]
# a second reference of the other class confirms pool[0]'s own class
y = cls.index("Y") if "Y" in cls else None
for i in range(n):
    back = t(i, y) if (y is not None and i != y) else float("nan")
    print("buffer %2d at %#014x (GiB %7.2f, %% 64 GiB = %5.2f)  as S' next to 0: %s ms -> class %s;  as A' next to %s: %.3f ms"
          % (i, pool[i].data_ptr(), pool[i].data_ptr() / 2**30, (pool[i].data_ptr() / 2**30) % 64, "%.3f" % ms[i] if ms[i] else "  -  ", cls[i], y, back))
