#!/bin/bash
for a in 1 2 6 7 9 10 12 13 14 15 17 18 19 20 21 22 23 24 25 26 27 28 29 30 31 33; do
  H2R_RECORD_PAD=$a timeout 100 python bench.py --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 > /tmp/ab.json
  python - <<PY
import json
d=json.load(open('/tmp/ab.json'))
print("pad", $a, "trace_ms", d["roofline"]["avg_launch_ms"], "GB/s", d["roofline"]["achieved"], "step_ms", d["ms_per_step"])
PY
done
