#!/bin/bash
# (the knobs are read by the -DH2R_DEV_KNOBS build only: python -m halo2_rsa_amd._build devknobs -DH2R_DEV_KNOBS; run with H2R_LIB=halo2_rsa_amd/lib/variants/devknobs.so)
# A/B for the Montgomery pipelined advice forms: one-wave row programs (H2R_ROWPROG_STAGE_ROWS=64) and the next call's chains ordered behind
# the previous call's cells kernel (H2R_ADV_CHAIN_AFTER_CELLS=1).
R=$GRAFT_REPO_ROOT; cd $R; export H2R_LIB=$R/halo2_rsa_amd/lib/variants/devknobs.so
for f in "--columns --montgomery" "--verify --columns --montgomery" "--montgomery"; do
  for cfg in "256 0" "64 0"; do   # (the explicit chain-after-cells order of the first A/B, H2R_ADV_CHAIN_AFTER_CELLS, was dropped from the library)
    set -- $cfg
    H2R_ROWPROG_STAGE_ROWS=$1 H2R_ADV_CHAIN_AFTER_CELLS=$2 timeout -s KILL 200 python bench.py --advice $f --sub-runs off --no-cpu-baseline --pmc-traffic off --steps 30 2>/dev/null | tail -1 | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print('%-36s SR=%-4s after=%s  %.4f ms/call  frac %.4f  kept %s  rows %.3f ms chain %.3f' % ('$f', '$1', '$2', d['ms_per_step'], d['roofline']['frac'], d['config']['buffer_placement']['kept_ms'], d['roofline']['in_field_rows_kernel_avg_ms'], d['roofline']['chain_kernel_avg_ms']))"
  done
done
