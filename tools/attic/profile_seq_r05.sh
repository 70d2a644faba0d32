# Per-kernel times of one advice call with nothing overlapped (bench.py --no-pipeline), per representation: what the pipelined call has to hide.
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/prof_seq; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
A="--steps 10 --warmup 3 --no-cpu-baseline --pmc-traffic off --sub-runs off --no-pipeline --placement-candidates 0"
i=0
for f in "" "--columns" "--montgomery" "--columns --montgomery"; do
  i=$((i+1))
  timeout -s KILL 150 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt_seq$i -o r -- python $R/bench.py --advice $f $A > $O/bench_$i.txt 2>&1
  echo "== $f" >> $O/summary.txt; head -12 /tmp/kt_seq$i/r_kernel_stats.csv | cut -c1-160 >> $O/summary.txt
done
cat $O/summary.txt
