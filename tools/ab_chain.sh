#!/bin/bash
# Same-box A/B of chain_kernel build variants.  Build them first (no GPU needed), e.g.
#   python -m halo2_rsa_amd._build v0m8 -DH2R_PROD_VARIANT=0 -DH2R_CHAIN_MINB=8
# then: gpurun -- 'bash tools/ab_chain.sh v0m8 v1m4 ...'
for rep in 1 2; do for v in "$@"; do
  export H2R_LIB=$PWD/halo2_rsa_amd/lib/variants/$v.so
  python tools/sweep.py H2R_TAG $v-serial --steps 40 --warmup 4 --no-pipeline
  python tools/sweep.py H2R_TAG $v-pipe --steps 40 --warmup 4
done; done
