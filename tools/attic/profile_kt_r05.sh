R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/prof_r05; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
A="--steps 20 --warmup 3 --no-cpu-baseline --pmc-traffic off"
timeout -s KILL 150 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt_advice -o r -- python $R/bench.py --advice $A > /dev/null 2>&1
timeout -s KILL 150 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt_advice_cm -o r -- python $R/bench.py --advice --columns --montgomery $A > /dev/null 2>&1
timeout -s KILL 150 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt_lookup -o r -- python $R/bench.py --lookup --steps 8 --warmup 2 --no-cpu-baseline --pmc-traffic off > /dev/null 2>&1
for t in advice advice_cm lookup; do cp /tmp/kt_$t/r_kernel_stats.csv $O/kernel_stats_$t.csv; done
cd $R; for t in advice advice_cm; do python tools/timed_region_stats.py /tmp/kt_$t 20 > $O/kernel_stats_${t}_timed.csv 2>/dev/null; done
head -6 $O/kernel_stats_*.csv
