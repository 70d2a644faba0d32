#!/bin/bash
# (the knobs are read by the -DH2R_DEV_KNOBS build only: python -m halo2_rsa_amd._build devknobs -DH2R_DEV_KNOBS; run with H2R_LIB=halo2_rsa_amd/lib/variants/devknobs.so)
# A/B of the row programs' stage size (H2R_ROWPROG_STAGE_ROWS = 256 | 128 | 64) under the pipelined advice forms: ms per call.
R=$GRAFT_REPO_ROOT; cd $R; export H2R_LIB=$R/halo2_rsa_amd/lib/variants/devknobs.so
for f in "--columns --montgomery" "--verify --columns --montgomery" "" "--verify"; do
  for sr in 256 128 64; do
    H2R_ROWPROG_STAGE_ROWS=$sr timeout -s KILL 200 python bench.py --advice $f --sub-runs off --no-cpu-baseline --pmc-traffic off --steps 30 2>/dev/null | tail -1 | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print('%-36s SR=%-4s %.4f ms/call  frac %.4f  kept %s  in_field_rows %.3f ms' % ('$f', '$sr', d['ms_per_step'], d['roofline']['frac'], d['config']['buffer_placement']['kept_ms'], d['roofline']['in_field_rows_kernel_avg_ms']))"
  done
done
