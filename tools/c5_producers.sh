#!/bin/bash
# BASELINE config 5 (256 chains of 3,072 dependent mul_mods per call): one producer against two (bench.py --producers 2: two pipelines on two
# streams, calls alternate -- the chain kernels of two calls side by side).  Plain allocations (four 50 GB trace regions do not leave room for the
# arena's search), same box, both forms.
cd $GRAFT_REPO_ROOT
W="--workload rsa2048_e2048bit --batch 256 --warmup 2 --placement-candidates 0"
for rep in 1 2; do
  python tools/sweep.py H2R_TAG producers1 $W --steps 8
  python tools/sweep.py H2R_TAG producers2 $W --steps 8 --producers 2
done
python tools/sweep.py H2R_TAG var-producers1 --workload rsa1024_e65537 --steps 40 --warmup 4
python tools/sweep.py H2R_TAG var-producers2 --workload rsa1024_e65537 --steps 40 --warmup 4 --producers 2
python tools/sweep.py H2R_TAG c2-producers2 --steps 20 --warmup 5 --producers 2
