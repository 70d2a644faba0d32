#!/usr/bin/env python3
"""h2r_image_arena_create in fresh processes, by region size and candidate count: which shapes survive (a runtime abort was seen with
5.4 GB regions)."""
import subprocess, sys, os
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
CODE = """
import sys, torch
sys.path.insert(0, %r)
import halo2_rsa_amd as H
chip = H.BigIntChip(64, 2048)
a = H.TraceArena.for_images(chip, int(sys.argv[1]), regions=2, candidates=int(sys.argv[2]))
print('ok', [round(t, 3) for t in a.measurements_ms])
a.close()
""" % ROOT
for size, cand in [(1 << 30, 8), (3 << 30, 8), ((4 << 30) - (1 << 20), 8), ((4 << 30) + (1 << 20), 4), (5368463360, 4), (5368463360, 8), (12 << 30, 4)]:
    res = []
    for rep in range(4):
        out = subprocess.run([sys.executable, "-c", CODE, str(size), str(cand)], capture_output=True, text=True, timeout=120)
        res.append("ok" if out.returncode == 0 else "rc%d %s" % (out.returncode, [l for l in out.stderr.splitlines() if "Memobj" in l][-1:][0][-40:] if "Memobj" in out.stderr else ""))
    print("region %.2f GB, %d candidates:" % (size / 2**30, cand), res, flush=True)
