#!/bin/bash
# same-box A/B of two builds of libh2r.so on the verifier paths: usage tools/ab_verify_libs.sh <nameA> <nameB> [reps]
cd $GRAFT_REPO_ROOT
for rep in $(seq 1 ${3:-2}); do
for v in $1 $2; do
  L=halo2_rsa_amd/lib/variants/$v.so
  echo "== $v"
  python tools/sweep.py H2R_LIB $L --verify --messages 128 --steps 40 --warmup 4
  python tools/sweep.py H2R_LIB $L --verify --steps 40 --warmup 4
  python tools/sweep.py H2R_LIB $L --steps 40 --warmup 4
done; done
