#!/usr/bin/env python3
"""Is the plain-allocation headline the two-class relation between the trace buffers that are IN FLIGHT TOGETHER?  (The two-queue form keeps two
record kernels running: the buffers of calls k and k + 1.)  N plain torch allocations of config 2's trace size are classified against buffer 0
by a concurrent pair of fills on two streams (different class ~6.9 TB/s, same ~5.3); then the pipelined modpow_public_key loop of bench.py runs
with `depth` buffer sets chosen by class ORDER: alternating (X Y X Y), all of one class, and allocation order.  python tools/trace_pair_class_probe.py"""
import os, random, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
import halo2_rsa_amd as H
from halo2_rsa_amd import _lib
sys.argv = [sys.argv[0]]
import bench

w, bits, e = 64, 2048, 65537
B = 1024
chip = H.BigIntChip(w, bits)
ns, xs, un, ux = bench.synth_inputs(w, bits, 0, B)
n, x = chip.assign_integer(un), chip.assign_integer(ux)
pl = chip.pow_fixed_layout(e)
tbytes = B * pl.elem_stride
NB = int(os.environ.get("NB", "12"))
bufs = [torch.empty(tbytes, dtype=torch.uint8, device="cuda") for _ in range(NB)]
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()

def pair(a, b, reps=4):
    for r in range(reps + 1):
        if r == 1:
            torch.cuda.synchronize(); t0 = time.perf_counter()
        with torch.cuda.stream(s1): a.fill_(r)
        with torch.cuda.stream(s2): b.fill_(r)
    torch.cuda.synchronize()
    return 2 * tbytes * reps / (time.perf_counter() - t0) / 1e12

r0 = [0.0] + [pair(bufs[0], bufs[j]) for j in range(1, NB)]
r1 = [pair(bufs[1], bufs[0]), 0.0] + [pair(bufs[1], bufs[j]) for j in range(2, NB)]
print("pair fill (two streams, torch fill_) with buffer 0, TB/s:", " ".join("%.2f" % v for v in r0[1:]))
print("pair fill with buffer 1, TB/s:                            ", " ".join("%.2f" % v for v in r1[:1] + r1[2:]), flush=True)
allv = r0[1:] + r1[:1] + r1[2:]
lo, hi = min(allv), max(allv)
if hi / lo < 1.12:
    print("one class only among these buffers"); sys.exit(0)
thr = (lo + hi) / 2
same01 = r0[1] < thr
cls = []
for j in range(NB):
    if j == 0: cls.append("X")
    elif j == 1: cls.append("X" if same01 else "Y")
    else:
        s0, s1_ = r0[j] < thr, r1[j] < thr          # slow with 0 / slow with 1 = same class as it
        cls.append("X" if s0 else ("Y" if (s1_ and not same01) else ("Y" if not s0 and same01 else "?")))
print("classes:", "".join(cls), flush=True)
X = [i for i, c in enumerate(cls) if c == "X"]; Y = [i for i, c in enumerate(cls) if c == "Y"]

def run(order, depth, side=2, steps=40, warm=60):
    pipe = H.Pipeline(chip, depth, side)
    wss = [torch.empty(chip.workspace_bytes(B, pl.num_mul_mods), dtype=torch.uint8, device="cuda") for _ in range(depth)]
    outs = [chip._new_limbs(B) for _ in range(depth)]
    sts = [torch.zeros(B, dtype=torch.uint8, device="cuda") for _ in range(depth)]
    infs = [torch.empty(B * chip.in_field_layout()[0], dtype=torch.uint8, device="cuda") for _ in range(depth)]
    k = 0
    def call():
        nonlocal k
        s = k % depth
        pipe.modpow_public_key(x, e, n, bufs[order[s]], wss[s], outs[s], sts[s], infs[s]); k += 1
    for _ in range(warm): call()
    pipe.join(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps): call()
    pipe.join(); torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    algo = B * (pl.stream_bytes + chip.in_field_layout()[1] + 2 * chip.num_limbs * 8)
    del pipe
    return B / dt / 1e6, algo / dt / 8e12

def show(name, order, depth):
    if any(i is None for i in order): print("%-44s not enough buffers of a class" % name); return
    vals = [run(order, depth) for _ in range(2)]
    print("%-44s buffers %s (%s): %s" % (name, order, "".join(cls[i] for i in order), "  ".join("%.2f M assigns/s (whole path %.3f)" % v for v in vals)), flush=True)

g = lambda L, i: L[i] if i < len(L) else None
show("depth 3, allocation order", [0, 1, 2], 3)
show("depth 3, X Y X", [g(X, 0), g(Y, 0), g(X, 1)], 3)
show("depth 3, one class (X X X)", [g(X, 0), g(X, 1), g(X, 2)], 3)
show("depth 3, one class (Y Y Y)", [g(Y, 0), g(Y, 1), g(Y, 2)], 3)
show("depth 4, X Y X Y", [g(X, 0), g(Y, 0), g(X, 1), g(Y, 1)], 4)
show("depth 4, X X Y Y", [g(X, 0), g(X, 1), g(Y, 0), g(Y, 1)], 4)
show("depth 4, one class (X X X X)", [g(X, 0), g(X, 1), g(X, 2), g(X, 3)], 4)
show("depth 4, allocation order", [0, 1, 2, 3], 4)
