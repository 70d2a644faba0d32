#!/bin/bash
# same-box A/B of the accumulator-plane layout
for rep in 1 2; do for lay in planar interleaved; do
  export H2R_ACC_LAYOUT=$lay
  python tools/sweep.py H2R_ACC_LAYOUT $lay --steps 40 --warmup 4 --no-pipeline
  python tools/sweep.py H2R_ACC_LAYOUT $lay --batch 8192 --steps 6 --warmup 2 --no-pipeline
  python tools/sweep.py H2R_ACC_LAYOUT $lay --workload rsa4096_w32_e65537 --batch 2048 --steps 5 --warmup 1 --no-pipeline
  python tools/sweep.py H2R_ACC_LAYOUT $lay --workload rsa1024_e65537 --steps 40 --warmup 4 --no-pipeline
done; done
