#!/bin/bash
# same-box A/B runs behind the round-3 step builds (developer builds with -DH2R_DEV_KNOBS; see profiles/r03_step_shapes.txt)
cd $GRAFT_REPO_ROOT
for lib in knobs knobs6; do
  export H2R_LIB=halo2_rsa_amd/lib/variants/$lib.so
  echo "== $lib: chain kernel register budget $( [ $lib = knobs6 ] && echo '6 waves/SIMD for K > 64' || echo '3 waves/SIMD for K > 64 (shipped)')"
  H2R_PIPE_STEP=0 python tools/sweep.py H2R_PIPE_STEP 0 --workload rsa3072_e65537 --steps 20 --warmup 3 --pmc-traffic off
  H2R_PIPE_STEP=0 python tools/sweep.py H2R_PIPE_STEP 0 --workload rsa4096_e65537 --steps 20 --warmup 3 --pmc-traffic off
  H2R_PIPE_STEP=0 python tools/sweep.py H2R_PIPE_STEP 0 --workload rsa4096_w32_e65537 --batch 4096 --steps 4 --warmup 1 --pmc-traffic off
done
export H2R_LIB=halo2_rsa_amd/lib/variants/knobs.so
for n in 2 3 4 6; do
  python tools/sweep.py H2R_STEP_CHAIN_X2_PER_CU $n --workload rsa4096_w32_e65537 --batch 4096 --steps 4 --warmup 1 --pmc-traffic off
  python tools/sweep.py H2R_STEP_CHAIN_X2_PER_CU $n --workload rsa4096_e65537 --steps 20 --warmup 3 --pmc-traffic off
done
for n in 3 4 6 8; do
  python tools/sweep.py H2R_STEP_CHAIN_X2_PER_CU $n --workload rsa3072_e65537 --steps 20 --warmup 3 --pmc-traffic off
done
