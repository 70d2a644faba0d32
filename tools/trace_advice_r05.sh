#!/bin/bash
# Kernel timeline of the pipelined advice forms (rocprofv3 --kernel-trace): who runs when inside one call.  argv: bench flags after --advice
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/trace_adv; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
tag=$1; shift
timeout -s KILL 150 rocprofv3 --kernel-trace --output-format csv -d /tmp/tr_$tag -o r -- python $R/bench.py --advice "$@" --steps 6 --warmup 3 --no-cpu-baseline --pmc-traffic off --sub-runs off --placement-candidates 4 > /dev/null 2>&1
f=$(find /tmp/tr_$tag -name "*kernel_trace.csv" | head -1)
python3 - "$f" > $O/timeline_$tag.txt <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
ev = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("void h2r::", "").replace("h2r::", "")[:40], r.get("Queue_Id", "?")) for r in rows]
ev.sort()
# the last three cells launches and everything between
cells = [i for i, e in enumerate(ev) if "cells_kernel" in e[2]]
lo = cells[-4] if len(cells) >= 4 else 0
t0 = ev[lo][0]
for s, e, n, q in ev[lo:]:
    print("%9.1f us  +%8.1f us  q%-3s %s" % ((s - t0) / 1e3, (e - s) / 1e3, q, n))
PY
tail -60 $O/timeline_$tag.txt
