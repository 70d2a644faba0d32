#!/bin/bash
for pr in 1 0; do
for a in 0 20000 32000 45000 60000 100000; do
  H2R_TRACE_PRIO=$pr H2R_TRACE_DYN_LDS=$a timeout 100 python bench.py --steps 40 --warmup 4 --no-cpu-baseline 2>/dev/null | tail -1 > /tmp/ab.json
  python - <<PY
import json
d=json.load(open('/tmp/ab.json'))
print("prio", $pr, "dyn_lds", $a, "trace_ms", d["roofline"]["avg_launch_ms"], "chain_ms", d["roofline"]["chain_kernel_avg_ms"], "step_ms", d["ms_per_step"], "value", d["value"])
PY
done; done
