#!/bin/bash
# same-box A/B of record-store cache policies: python -m halo2_rsa_amd._build plain -DH2R_STORE_PLAIN; python -m halo2_rsa_amd._build nt; then gpurun -- bash tools/ab_nt.sh plain nt
for rep in 1 2; do for v in "$@"; do
  export H2R_LIB=$PWD/halo2_rsa_amd/lib/variants/$v.so
  python tools/sweep.py H2R_TAG $v-c2pipe --steps 40 --warmup 4
  python tools/sweep.py H2R_TAG $v-c2serial --steps 40 --warmup 4 --no-pipeline
  python tools/sweep.py H2R_TAG $v-8kserial --batch 8192 --steps 6 --warmup 2 --no-pipeline
done; done
