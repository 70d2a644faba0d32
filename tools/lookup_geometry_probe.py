#!/usr/bin/env python3
"""lookup_fill_kernel's work distribution against the placement classes, on the SAME buffers: every partner of pool[0] as S', the call timed under
each geometry -- rows of a column per workgroup (8,192 shipped ... 256) x contiguous runs / round-robin 256-row blocks among the column's workgroups
(developer build: the geometry rides in bits 8-16 of arg_mask).  H2R_LIB=halo2_rsa_amd/lib/variants/devknobs.so python tools/lookup_geometry_probe.py"""
import os, random, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
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
hist = la.new_hist(B); la.hist_records(res.trace, hist); torch.cuda.synchronize(); del res; torch.cuda.empty_cache()
P = 21888242871839275222246405745257275088548364400416034343698204186575808495617
thetas = [rng.randrange(P) for _ in range(B)]
n = int(os.environ.get("NB", "14"))
pool = [torch.empty((B, 5, usable, 32), dtype=torch.uint8, device="cuda") for _ in range(n)]
ref = None
def t(ia, is_, mask):
    la.permuted_columns(hist, thetas, usable, arg_mask=mask, out=(pool[ia], pool[is_]))
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(3): la.permuted_columns(hist, thetas, usable, arg_mask=mask, out=(pool[ia], pool[is_]))
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / 3
geos = [(8192, 0), (8192, 1), (2048, 0), (2048, 1), (1024, 0), (1024, 1), (512, 1), (256, 0), (256, 1)]
print("partner  " + "  ".join("%5d%s" % (r, "rr" if rr else "  ") for r, rr in geos) + "   (ms per call; rr = round-robin blocks)")
rows = []
for i in range(1, n):
    v = [t(0, i, 31 | ((r // 256) << 8) | (rr << 16)) for r, rr in geos]
    rows.append(v)
    print("0 + %-3d  " % i + "  ".join("%7.3f" % x for x in v), flush=True)
print("slowest  " + "  ".join("%7.3f" % max(r[k] for r in rows) for k in range(len(geos))))
print("fastest  " + "  ".join("%7.3f" % min(r[k] for r in rows) for k in range(len(geos))))
# the columns must not depend on the geometry
la.permuted_columns(hist, thetas, usable, arg_mask=31, out=(pool[0], pool[1]))
la.permuted_columns(hist, thetas, usable, arg_mask=31 | (1 << 8) | (1 << 16), out=(pool[2], pool[3]))
torch.cuda.synchronize()
print("byte-identical columns under (8192, contiguous) and (256, round-robin):", bool(torch.equal(pool[0], pool[2]) and torch.equal(pool[1], pool[3])))
