#!/bin/bash
# Round 6: a kernel trace of the form the driver's command runs (bench.py --gpus 1 --steps 20 --warmup 5 -> two-queue form, trace_kernel<64,32>).
# First with the product library (does the profiler flip the queue probe?), then -- if it did -- with the developer build forcing the form
# (python -m halo2_rsa_amd._build devknobs -DH2R_DEV_KNOBS; H2R_PIPE_FORM=1).  gpurun --timeout 900 -- 'bash tools/profile_two_queue_r06.sh'
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(cd "$(dirname "$0")/.." && pwd)
O=$R/gpurun_out/prof_r06_2q; rm -rf $O; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
T="timeout -s KILL"
ARGS="--gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --pmc-traffic off --sub-runs off --scale-anchor off"
$T 200 python $R/bench.py $ARGS > $O/bench_unprofiled.json 2> $O/bench_unprofiled.err
$T 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -o r -- python $R/bench.py $ARGS > $O/kt.log 2>&1
H2R_LIB=$R/halo2_rsa_amd/lib/variants/devknobs.so H2R_PIPE_FORM=1 $T 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_forced -o r -- python $R/bench.py $ARGS > $O/kt_forced.log 2>&1
H2R_LIB=$R/halo2_rsa_amd/lib/variants/devknobs.so H2R_PIPE_FORM=1 $T 200 python $R/bench.py $ARGS > $O/bench_forced_unprofiled.json 2>/dev/null
cd $R
for k in kt kt_forced; do
  python tools/two_queue_profile.py $O/$k 20 1251760128 > $O/two_queue_$k.csv 2>$O/two_queue_$k.err
  python tools/timed_region_stats.py $O/$k 20 > $O/kernel_stats_$k.csv 2>/dev/null
  grep -o '"pipeline_form": {[^}]*}' $O/$k.log | head -1 > $O/form_$k.txt
  find $O/$k -name "*kernel_trace.csv" -size +20M -delete
done
grep -o '"pipeline_form": {[^}]*}' $O/bench_unprofiled.json | head -1
cat $O/form_kt.txt $O/form_kt_forced.txt
tail -8 $O/two_queue_kt.csv; tail -8 $O/two_queue_kt_forced.csv
grep -o '"value": [0-9.]*' $O/bench_unprofiled.json $O/bench_forced_unprofiled.json $O/kt.log $O/kt_forced.log
rm -rf $O/kt/*/*agent* $O/kt_forced/*/*agent*
du -sh $O
