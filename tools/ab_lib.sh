#!/bin/bash
# same-box A/B of two builds of libh2r.so (halo2_rsa_amd/lib/variants/<name>.so): usage tools/ab_lib.sh <nameA> <nameB> [reps]
cd $GRAFT_REPO_ROOT
for rep in $(seq 1 ${3:-2}); do
for v in $1 $2; do
  L=halo2_rsa_amd/lib/variants/$v.so
  echo "== $v"
  python tools/sweep.py H2R_LIB $L --no-pipeline --steps 30 --warmup 5 --placement-candidates 0
  python tools/sweep.py H2R_LIB $L --steps 20 --warmup 5
  python tools/sweep.py H2R_LIB $L --workload rsa2048_e2048bit --batch 256 --steps 6 --warmup 2
  python tools/sweep.py H2R_LIB $L --workload rsa1024_e65537 --steps 40 --warmup 4
  python tools/sweep.py H2R_LIB $L --workload rsa3072_e65537 --steps 20 --warmup 3
  python tools/sweep.py H2R_LIB $L --workload rsa4096_e65537 --steps 20 --warmup 3
done; done
