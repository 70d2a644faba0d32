#!/bin/bash
# BASELINE config 5 (RSA-2048, 2048-bit exponent, batch 256): chain waves per signature x pipeline shape
W="--workload rsa2048_e2048bit --batch 256 --steps 6 --warmup 2"
for nw in 4 8; do
  export H2R_CHAIN_NW=$nw
  python tools/sweep.py H2R_TAG nw$nw-serial $W --no-pipeline
  python tools/sweep.py H2R_TAG nw$nw-d2s1 $W
  python tools/sweep.py H2R_TAG nw$nw-d3s2 $W --pipeline-depth 3 --side-streams 2
done
