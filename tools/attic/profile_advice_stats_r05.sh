#!/bin/bash
# rocprofv3 --kernel-trace --stats of the pipelined advice forms as shipped at the end of round 5 -> profiles/r05_kernel_stats_advice_*.csv
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/prof_adv; rm -rf $O; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
A="--steps 20 --warmup 3 --no-cpu-baseline --pmc-traffic off --sub-runs off"
run() { tag=$1; shift
  timeout -s KILL 150 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt_$tag -o r -- python $R/bench.py --advice "$@" $A > $O/bench_$tag.txt 2>&1
  cp /tmp/kt_$tag/r_kernel_stats.csv $O/kernel_stats_$tag.csv
  (cd $R; python tools/timed_region_stats.py /tmp/kt_$tag 20 > $O/kernel_stats_${tag}_timed.csv 2>/dev/null)
}
run advice
run advice_cm --columns --montgomery
run advice_verify --verify
run advice_verify_cm --verify --columns --montgomery
for t in advice advice_cm advice_verify advice_verify_cm; do echo == $t; sed -n 3,6p $O/kernel_stats_${t}_timed.csv | cut -c1-150; tail -1 $O/bench_$t.txt | cut -c1-120; done
