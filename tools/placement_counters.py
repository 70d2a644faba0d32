"""Placement classes under hardware counters.

Finds, by timing, one FAST and one SLOW placement of the SAME size for two kernels, then launches them alternately at the END of the
process so that a `rocprofv3 --kernel-trace --pmc ...` run of this script can tell the classes apart by counter:

  lookup_fill_kernel   A' = pool[0], S' = a buffer of the other class (fast) / of pool[0]'s class (slow)   (tools/lookup_class_probe.py)
  trace_kernel         the record kernel alone (batch 512: one chain + one record kernel, no internal overlap) into the fastest / slowest
                       of N plain allocations of the trace size

The last line printed is `SEQ {...}`: per kernel the labels (F / S) of its last dispatches in launch order, and the timings the choice
was made on.  tools/placement_counters_parse.py joins it with the counter CSVs."""
import json
import os
import random
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import halo2_rsa_amd as H
from halo2_rsa_amd import _lib

REPS = int(os.environ.get("PC_REPS", "3"))
N_LOOK = int(os.environ.get("PC_N_LOOK", "10"))
N_TRACE = int(os.environ.get("PC_N_TRACE", "24"))
TRACE_BATCH = int(os.environ.get("PC_TRACE_BATCH", "512"))
out = {}
rng = random.Random(1)
chip = H.BigIntChip(64, 2048)

# ---- lookup columns ------------------------------------------------------------------------------------------------
B = 256
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
pool = [torch.empty((B, 5, usable, 32), dtype=torch.uint8, device="cuda") for _ in range(N_LOOK)]


def look(ia, is_, reps=2):
    la.permuted_columns(hist, thetas, usable, out=(pool[ia], pool[is_]))
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        la.permuted_columns(hist, thetas, usable, out=(pool[ia], pool[is_]))
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps


ms = [None] + [look(0, i) for i in range(1, N_LOOK)]
fast_i = min(range(1, N_LOOK), key=lambda i: ms[i])
slow_i = max(range(1, N_LOOK), key=lambda i: ms[i])
print("lookup: as S' next to pool[0] (ms):", " ".join("%.3f" % m for m in ms[1:]), "-> fast partner", fast_i, "slow partner", slow_i, flush=True)
for i in range(N_LOOK):
    print("  pool[%d] at %#x" % (i, pool[i].data_ptr()))
out["lookup"] = {"ms_by_partner": ms[1:], "fast": fast_i, "slow": slow_i, "bytes_per_call": 2 * B * 5 * usable * 32,
                 "addr": [pool[i].data_ptr() for i in range(N_LOOK)]}

# ---- trace regions ---------------------------------------------------------------------------------------------------
TB = TRACE_BATCH
Nt = [rng.getrandbits(2048) | (1 << 2047) | 1 for _ in range(TB)]
Xt = [rng.randrange(n) for n in Nt]
xa, na = chip.assign_integer(Xt), chip.assign_integer(Nt)
pl = chip.pow_fixed_layout(65537)
tbytes = TB * pl.elem_stride
# the lookup pool stays allocated while the trace buffers are (they must not simply reuse one freed block)
tpool = [torch.empty(tbytes, dtype=torch.uint8, device="cuda") for _ in range(N_TRACE)]
ws = torch.empty(chip.workspace_bytes(TB, pl.num_mul_mods), dtype=torch.uint8, device="cuda")
outl = chip._new_limbs(TB)
st = torch.zeros(TB, dtype=torch.uint8, device="cuda")


def rec(i):
    chip.pow_mod_fixed_exp(xa, 65537, na, trace_buf=tpool[i], workspace=ws, out=outl, status=st)


for i in range(N_TRACE):
    rec(i)
torch.cuda.synchronize()
_lib.profile_enable(4096)
for _ in range(3):
    for i in range(N_TRACE):
        rec(i)
torch.cuda.synchronize()
tms = _lib.profile_read(1)   # H2R_KERNEL_TRACE
_lib.profile_enable(0)
per = [sum(tms[i + N_TRACE * r] for r in range(3)) / 3 for i in range(N_TRACE)]
tf = min(range(N_TRACE), key=lambda i: per[i])
ts = max(range(N_TRACE), key=lambda i: per[i])
print("trace: record kernel alone per buffer (ms):", " ".join("%.4f" % m for m in per), "-> fastest", tf, "slowest", ts, flush=True)
out["trace"] = {"ms_by_buffer": per, "fast": tf, "slow": ts, "bytes_per_launch": TB * pl.num_mul_mods * 64338,
                "batch": TB, "addr": [t.data_ptr() for t in tpool]}

# ---- the labelled dispatches (LAST in the process) ----------------------------------------------------------------------
seq_l, seq_t = [], []
torch.cuda.synchronize()
for _ in range(REPS):
    look(0, fast_i, reps=1); seq_l += ["F", "F"]     # look() = one untimed + `reps` timed calls, each ONE lookup_fill_kernel dispatch
    look(0, slow_i, reps=1); seq_l += ["S", "S"]
for _ in range(REPS):
    rec(tf); seq_t.append("F")
    rec(ts); seq_t.append("S")
torch.cuda.synchronize()
out["seq"] = {"lookup_fill_kernel": seq_l, "trace_kernel": seq_t}
print("SEQ " + json.dumps(out))
