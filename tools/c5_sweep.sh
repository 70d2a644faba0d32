#!/bin/bash
# BASELINE config 5 (RSA-2048, 2048-bit exponent, batch 256): chain build (waves per signature x latency build) x pipeline
W="--workload rsa2048_e2048bit --batch 256 --steps 5 --warmup 2"
for cfg in "4 0" "4 1" "8 0" "8 1"; do set -- $cfg
  export H2R_CHAIN_NW=$1 H2R_CHAIN_DEEP=$2
  python tools/sweep.py H2R_TAG nw$1-deep$2-serial $W --no-pipeline
  python tools/sweep.py H2R_TAG nw$1-deep$2-d2s1 $W
done
