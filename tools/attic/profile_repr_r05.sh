#!/bin/bash
# Round 5, after the Montgomery kernel's LDS diet (six waves per CU): the representation profile set again.
# gpurun_out/prof_repr -> profiles/r05_cells_representations.txt, r05_bench_advice_*montgomery*.json, r05_kernel_stats_advice_cm*.csv
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/prof_repr; rm -rf $O; mkdir -p $O; cd $R
{
for r in 0 2 1 3; do timeout -s KILL 120 tools/_bin/cells_repr_bench 64 2048 1024 $r; done
timeout -s KILL 120 tools/_bin/cells_repr_bench 32 4096 256 3
timeout -s KILL 120 tools/_bin/cells_repr_bench 64 4096 256 3
} > $O/cells_representations.txt 2>&1
timeout -s KILL 300 python bench.py --advice --montgomery --sub-runs off 2>/dev/null | tail -1 > $O/bench_advice_montgomery.json
timeout -s KILL 300 python bench.py --advice --columns --montgomery --sub-runs off 2>/dev/null | tail -1 > $O/bench_advice_columns_montgomery.json
cd /tmp; export TMPDIR=/tmp
A="--steps 20 --warmup 3 --no-cpu-baseline --pmc-traffic off --sub-runs off"
timeout -s KILL 150 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt_advice_cm -o r -- python $R/bench.py --advice --columns --montgomery $A > /dev/null 2>&1
cp /tmp/kt_advice_cm/r_kernel_stats.csv $O/kernel_stats_advice_cm.csv
cd $R; python tools/timed_region_stats.py /tmp/kt_advice_cm 20 > $O/kernel_stats_advice_cm_timed.csv 2>/dev/null
head -4 $O/kernel_stats_advice_cm_timed.csv; grep -E "representation|full  " $O/cells_representations.txt
