#!/bin/bash
# Round 5: the headline path's profile set (trace / step kernels), same recipe as tools/profile_round.sh, every command under a hard timeout.
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/prof_r05h; rm -rf $O; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
T="timeout -s KILL"
ARGS="--steps 20 --warmup 3 --no-cpu-baseline --pmc-traffic off --sub-runs off --scale-anchor off"
$T 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -o r -- python $R/bench.py $ARGS > $O/kt.log 2>&1
$T 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_serial -o r -- python $R/bench.py $ARGS --no-pipeline > $O/kt_serial.log 2>&1
$T 200 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmc_w -o r -- python $R/bench.py --steps 5 --warmup 1 --no-cpu-baseline --pmc-traffic off --sub-runs off --scale-anchor off --no-pipeline > $O/pmc_w.log 2>&1
$T 200 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_r -o r -- python $R/bench.py --steps 5 --warmup 1 --no-cpu-baseline --pmc-traffic off --sub-runs off --scale-anchor off --no-pipeline > $O/pmc_r.log 2>&1
$T 200 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmc_step_w -o r -- python $R/bench.py --steps 5 --warmup 1 --no-cpu-baseline --pmc-traffic off --sub-runs off --scale-anchor off --pipeline-depth 2 --side-streams 1 > $O/pmc_step_w.log 2>&1
$T 200 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_step_r -o r -- python $R/bench.py --steps 5 --warmup 1 --no-cpu-baseline --pmc-traffic off --sub-runs off --scale-anchor off --pipeline-depth 2 --side-streams 1 > $O/pmc_step_r.log 2>&1
cd $R
python tools/pmc_to_json.py $O 1024 > $O/pmc_traffic_headline.json 2>$O/pmc_to_json.err
python tools/timeline.py $O/kt > $O/timeline_pipeline.txt 2>/dev/null
python tools/timed_region_stats.py $O/kt 20 > $O/kernel_stats_pipeline_timed.csv 2>/dev/null
python tools/timed_region_stats.py $O/kt_serial 20 > $O/kernel_stats_serial_timed.csv 2>/dev/null
cp $O/kt/r_kernel_stats.csv $O/kernel_stats_pipeline.csv; cp $O/kt_serial/r_kernel_stats.csv $O/kernel_stats_serial.csv
$T 300 python bench.py --steps 40 --warmup 4 --sub-runs off > $O/bench_pipeline.json 2> $O/bench_pipeline.err
$T 200 python bench.py --steps 40 --warmup 4 --no-pipeline --no-cpu-baseline --pmc-traffic off --sub-runs off > $O/bench_serial.json 2>/dev/null
$T 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --pmc-traffic off --placement-candidates 0 > $O/bench_pipeline_driver_args_plain_allocations.json 2>/dev/null
$T 200 python bench.py --steps 40 --warmup 4 --verify --no-cpu-baseline --pmc-traffic off --sub-runs off > $O/bench_verify.json 2>/dev/null
H2R_FORCE_DIST=1 $T 200 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29611 bench.py --gpus 1 --batch 2048 --chunks 4 --steps 20 --warmup 5 --no-cpu-baseline --pmc-traffic off > $O/bench_torchrun1_config3_shard.json 2> $O/bench_torchrun1.err
GPU_MAX_HW_QUEUES=4 H2R_FORCE_DIST=1 $T 200 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29612 bench.py --gpus 1 --batch 2048 --chunks 4 --steps 20 --warmup 5 --no-cpu-baseline --pmc-traffic off > $O/bench_torchrun1_config3_shard_default_queues.json 2>/dev/null
{
  echo "# other BASELINE configs and shapes, same box (tools/sweep.py lines)"
  for cfg in "C3-shard-8192-as-4-calls --batch 2048 --chunks 4 --steps 20 --warmup 5" "C3-shard-8192-one-call --batch 8192 --steps 6 --warmup 2" \
             "C4-rsa4096-w32-4096 --workload rsa4096_w32_e65537 --batch 4096 --steps 8 --warmup 2" "C5-e2048bit-256 --workload rsa2048_e2048bit --batch 256 --steps 8 --warmup 2" \
             "rsa1024 --workload rsa1024_e65537 --steps 40 --warmup 4" "rsa3072 --workload rsa3072_e65537 --steps 20 --warmup 3" "rsa4096-w64 --workload rsa4096_e65537 --steps 20 --warmup 3" \
             "rsa2048-shared-modulus --steps 40 --warmup 4 --shared-modulus" "C1-one-signature-per-call --batch 1 --steps 200 --warmup 20 --no-pipeline"; do
    $T 150 python tools/sweep.py CONFIG $cfg
  done
} > $O/other_configs.txt 2>&1
$T 150 python tools/var_exponent_timing.py > $O/var_exponent.txt 2>&1
$T 150 python tools/lookup_timing.py > $O/lookup_timing.txt 2>&1
$T 150 python tools/lookup_placement_probe.py 6 > $O/lookup_placement.txt 2>&1
rm -rf $O/kt $O/kt_serial $O/pmc_w $O/pmc_r $O/pmc_step_w $O/pmc_step_r
ls -la $O; tail -1 $O/bench_pipeline.json | cut -c1-400; cat $O/other_configs.txt | tail -12
