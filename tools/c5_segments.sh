#!/bin/bash
# BASELINE config 5 and the Var path with a long exponent walked as segments of its bits (exp_segment_count): pipelined and single calls
cd $GRAFT_REPO_ROOT
W="--workload rsa2048_e2048bit --batch 256"
python tools/sweep.py H2R_TAG c5-pipelined-8steps $W --steps 8 --warmup 2
python tools/sweep.py H2R_TAG c5-pipelined-20steps $W --steps 20 --warmup 2
python tools/sweep.py H2R_TAG c5-serial $W --steps 4 --warmup 1 --no-pipeline
python tools/sweep.py H2R_TAG c5-pipelined-8steps $W --steps 8 --warmup 2
python tools/sweep.py H2R_TAG c2-driver-args --steps 20 --warmup 5
