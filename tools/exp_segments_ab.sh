#!/bin/bash
# segments a long exponent is walked in (developer build -DH2R_DEV_KNOBS, H2R_EXP_SEGMENTS): BASELINE config 5 pipelined and as single calls, same box
cd $GRAFT_REPO_ROOT
export H2R_LIB=halo2_rsa_amd/lib/variants/knobs.so
W="--workload rsa2048_e2048bit --batch 256"
for s in 1 2 4 8 16 32; do
  python tools/sweep.py H2R_EXP_SEGMENTS $s $W --steps 8 --warmup 2
  python tools/sweep.py H2R_EXP_SEGMENTS $s $W --steps 4 --warmup 1 --no-pipeline
done
