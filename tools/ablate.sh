#!/bin/bash
# timing experiments: H2R_ABLATE bit0 = skip carry phase, bit1 = skip product loop (outputs are then wrong)
# needs the developer build:  python -m halo2_rsa_amd._build ablation -DH2R_ABLATION   (run before gpurun)
export H2R_LIB=$PWD/halo2_rsa_amd/lib/variants/ablation.so
for a in 0 4 8 12 16; do
  H2R_ABLATE=$a timeout 100 python bench.py --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 > /tmp/ab.json
  python - <<PY
import json
d=json.load(open('/tmp/ab.json'))
print("ablate", $a, "trace_ms", d["roofline"]["avg_launch_ms"], "chain_ms", d["roofline"]["chain_kernel_avg_ms"], "step_ms", d["ms_per_step"])
PY
done
