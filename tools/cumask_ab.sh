#!/bin/bash
# Developer experiment: the pipeline's record stream created with a CU mask (H2R_PIPE_CU_MASK, -DH2R_DEV_KNOBS build as lib/variants/knobs.so).
# hipExtStreamCreateWithCUMask makes a BLOCKING stream (it synchronises with the null stream), so the caller must be on a stream of its own.
export H2R_LIB=$PWD/halo2_rsa_amd/lib/variants/knobs.so
for rep in 1 2; do
python tools/sweep.py H2R_PIPE_CU_MASK 0,55555555,0000ffff,77777777,33333333,ffffffff --steps 40 --warmup 4 --user-stream 2>&1 | grep "H2R_PIPE"
done
