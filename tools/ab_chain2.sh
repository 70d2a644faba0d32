#!/bin/bash
# Same-box A/B of chain_kernel build variants over the shapes where the chain kernel's own time shows:
#   gpurun -- 'bash tools/ab_chain2.sh v1 v2 ...'   (variants built with python -m halo2_rsa_amd._build <name> -DFLAG)
for rep in 1 2; do for v in "$@"; do
  export H2R_LIB=$PWD/halo2_rsa_amd/lib/variants/$v.so
  python tools/sweep.py H2R_TAG $v-serial-shared --steps 40 --warmup 4 --no-pipeline --shared-modulus 2>&1 | grep H2R_TAG
  python tools/sweep.py H2R_TAG $v-serial --steps 40 --warmup 4 --no-pipeline 2>&1 | grep H2R_TAG
  python tools/sweep.py H2R_TAG $v-pipe --steps 40 --warmup 4 2>&1 | grep H2R_TAG
  python tools/sweep.py H2R_TAG $v-c5 --workload rsa2048_e2048bit --batch 256 --steps 4 --warmup 1 2>&1 | grep H2R_TAG
  python tools/sweep.py H2R_TAG $v-3072 --workload rsa3072_e65537 --steps 20 --warmup 3 2>&1 | grep H2R_TAG
done; done
