#!/bin/bash
for a in 0 1 2 3 5 7 9 13; do
  H2R_PLANE_PAD=$a timeout 100 python bench.py --steps 40 --warmup 4 --no-cpu-baseline --no-pipeline 2>/dev/null | tail -1 > /tmp/ab.json
  python - <<PY
import json
d=json.load(open('/tmp/ab.json'))
print("plane_pad", $a, "trace_ms", d["roofline"]["avg_launch_ms"], "GB/s", d["roofline"]["achieved"], "step_ms", d["ms_per_step"])
PY
done
