#!/bin/bash
# [r6] RSA-1024 with the one-wave chain: the one-launch step against the two-queue form, by signatures per call (profiles/r06_two_queue_rsa1024.txt).
# Developer build (python -m halo2_rsa_amd._build devknobs -DH2R_DEV_KNOBS): H2R_PIPE_TWOQ_L16 = -1 never | n = for every call of up to n | unset = the shipped rule.
cd $GRAFT_REPO_ROOT
export H2R_LIB=$PWD/halo2_rsa_amd/lib/variants/devknobs.so
A="--workload rsa1024_e65537 --steps 20 --warmup 5 --no-cpu-baseline --pmc-traffic off --sub-runs off --scale-anchor off"
line() { python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('%.2f M  whole %.3f  frac %.3f  %s' % (d['value']/1e6, d.get('whole_path_hbm_frac') or 0, d['roofline']['frac'], (d['config'].get('pipeline_form') or {}).get('record_form')))"; }
for rep in 1 2; do
for B in 1024 1536 2048 4096 8192; do
echo "B=$B step:     $(H2R_PIPE_TWOQ_L16=-1 python bench.py $A --batch $B 2>/dev/null | line)"
echo "B=$B twoq:     $(H2R_PIPE_TWOQ_L16=8192 python bench.py $A --batch $B 2>/dev/null | line)"
echo "B=$B shipped:  $(python bench.py $A --batch $B 2>/dev/null | line)"
done
echo "4 x 2048 step:     $(H2R_PIPE_TWOQ_L16=-1 python bench.py $A --batch 2048 --chunks 4 2>/dev/null | line)"
echo "4 x 2048 shipped:  $(python bench.py $A --batch 2048 --chunks 4 2>/dev/null | line)"
done
