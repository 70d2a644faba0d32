#!/usr/bin/env python3
"""Round 3: on a box where the arena's candidates are ALL of the slow class (about one box in five), does any other way of
obtaining the memory give a fast region?  First a quick look (12 candidates of 256 MB chunks); if one is fast, exit.  Otherwise:
chunk sizes 2 / 16 / 64 MB / 1 GB (developer build, H2R_ARENA_CHUNK_MB), candidates behind 64 / 128 / 192 GB placeholders, torch.
usage: no_fast_box_probe.py [stage]   (stage is set by the script itself for its sub-processes)"""
import os, subprocess, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
stage = sys.argv[1] if len(sys.argv) > 1 else "main"
import torch
import halo2_rsa_amd as H
B = 1024
chip = H.BigIntChip(64, 2048)

def look(name, cands=12, behind_gb=0):
    dummy = torch.empty(behind_gb << 30, dtype=torch.uint8, device="cuda") if behind_gb else None
    a = H.TraceArena.for_pow(chip, 65537, B, regions=1, candidates=cands)
    v = sorted(a.measurements_ms)
    a.close()
    del dummy
    torch.cuda.empty_cache()
    print("%-40s best %.4f second %.4f median %.4f worst %.4f (%d of %d below 0.19)" % (name, v[0], v[1], v[len(v) // 2], v[-1], sum(t < 0.19 for t in v), len(v)), flush=True)
    return v[0]

if stage == "main":
    if look("arena, 256 MB chunks") < 0.19:
        print("this box has fast regions: nothing to probe")
        sys.exit(0)
    print("NO FAST REGION among the first candidates: probing alternatives", flush=True)
    for gb in (64, 128, 192):
        look("arena behind %d GB" % gb, behind_gb=gb)
    env = dict(os.environ, H2R_LIB=os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "halo2_rsa_amd", "lib", "variants", "knobs.so"))
    for mb in (2, 16, 64, 1024):
        env["H2R_ARENA_CHUNK_MB"] = str(mb)
        subprocess.run([sys.executable, os.path.abspath(__file__), "chunk%d" % mb], env=env)
else:
    look("arena, %s MB chunks" % stage[5:], cands=16)
