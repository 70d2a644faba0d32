#!/bin/bash
for nw in 8 4 2; do for m in "--no-pipeline" ""; do
  H2R_CHAIN_NW=$nw H2R_TRACE_PRIO=1 H2R_TRACE_DYN_LDS=60000 timeout 100 python bench.py --steps 40 --warmup 4 --no-cpu-baseline $m 2>/dev/null | tail -1 > /tmp/ab.json
  python - <<PY
import json
d=json.load(open('/tmp/ab.json'))
print("nw", $nw, "mode", "$m" or "pipeline", "trace_ms", d["roofline"]["avg_launch_ms"], "chain_ms", d["roofline"]["chain_kernel_avg_ms"], "step_ms", d["ms_per_step"], "value", d["value"])
PY
done; done
