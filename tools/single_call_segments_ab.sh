#!/bin/bash
# a single stream-ordered RSA-2048 call of 1,024 signatures (e = 65537) walked as segments of its 17 exponent bits (developer build, H2R_SINGLE_CALL_SEGMENTS)
cd $GRAFT_REPO_ROOT
export H2R_LIB=halo2_rsa_amd/lib/variants/knobs.so
for rep in 1 2; do
python tools/sweep.py H2R_SINGLE_CALL_SEGMENTS 1,2,3,4,6,8 --no-pipeline --steps 30 --warmup 5
done
python tools/sweep.py H2R_SINGLE_CALL_SEGMENTS 1,2,3,4 --no-pipeline --steps 30 --warmup 5 --batch 768
python tools/sweep.py H2R_SINGLE_CALL_SEGMENTS 1,2,3,4 --no-pipeline --steps 30 --warmup 5 --batch 1536
