#!/bin/bash
# Collect the per-round profile set on the GPU box into gpurun_out/prof_final (copied to profiles/ afterwards).
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/prof_final; rm -rf $O; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
ARGS="--steps 20 --warmup 3 --no-cpu-baseline --pmc-traffic off"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -o r -- python $R/bench.py $ARGS > $O/kt.log 2>&1
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_serial -o r -- python $R/bench.py $ARGS --no-pipeline > $O/kt_serial.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmc_w -o r -- python $R/bench.py --steps 5 --warmup 1 --no-cpu-baseline --pmc-traffic off --no-pipeline > $O/pmc_w.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_r -o r -- python $R/bench.py --steps 5 --warmup 1 --no-cpu-baseline --pmc-traffic off --no-pipeline > $O/pmc_r.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmc_step_w -o r -- python $R/bench.py --steps 5 --warmup 1 --no-cpu-baseline --pmc-traffic off --pipeline-depth 2 --side-streams 1 > $O/pmc_step_w.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_step_r -o r -- python $R/bench.py --steps 5 --warmup 1 --no-cpu-baseline --pmc-traffic off --pipeline-depth 2 --side-streams 1 > $O/pmc_step_r.log 2>&1
# the device-side flatten and the advice image of a config-2 trace: kernel stats + PMC passes
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_emit -o r -- python $R/tools/emit_timing.py 1024 rsa2048 > $O/kt_emit.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmc_emit_w -o r -- python $R/tools/emit_timing.py 1024 rsa2048 0 noadvice > /dev/null 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_emit_r -o r -- python $R/tools/emit_timing.py 1024 rsa2048 0 noadvice > /dev/null 2>&1
cd $R
python tools/pmc_to_json.py $O 1024 > $O/pmc_traffic.json
python tools/timeline.py $O/kt > $O/timeline_pipeline.txt
python tools/timed_region_stats.py $O/kt 20 > $O/kernel_stats_pipeline_timed.csv
python tools/timed_region_stats.py $O/kt_serial 20 > $O/kernel_stats_serial_timed.csv
timeout 600 python bench.py --steps 40 --warmup 4 > $O/bench_pipeline.json 2> $O/bench_pipeline.err
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_pipeline_driver_args.json 2>/dev/null
timeout 300 python bench.py --steps 40 --warmup 4 --no-pipeline --no-cpu-baseline --pmc-traffic off > $O/bench_serial.json 2>/dev/null
# the same box with plain allocations (no placement-aware arena): what the arena is worth
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --pmc-traffic off --placement-candidates 0 > $O/bench_pipeline_driver_args_plain_allocations.json 2>/dev/null
timeout 300 python bench.py --steps 40 --warmup 4 --pipeline-depth 3 --side-streams 2 --no-cpu-baseline --pmc-traffic off > $O/bench_pipeline_d3s2.json 2>/dev/null
timeout 300 python bench.py --steps 40 --warmup 4 --verify --no-cpu-baseline --pmc-traffic off > $O/bench_verify.json 2>/dev/null
# BASELINE config 3 as ONE rank sees it (8,192-signature shard = four pipelined calls of 2,048 per step, the --gpus N > 1 default), through torchrun + RCCL
H2R_FORCE_DIST=1 timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29611 bench.py --gpus 1 --batch 2048 --chunks 4 --steps 20 --warmup 5 --no-cpu-baseline --pmc-traffic off > $O/bench_torchrun1_config3_shard.json 2> $O/bench_torchrun1.err
{
  echo "# other BASELINE configs and shapes, same box (tools/sweep.py lines: step, value = assigns/s, record kernel, chain kernel)"
  echo "# (pipelined RSA-2048 and RSA-1024 calls of more than 512 signatures are issued as one-launch steps (parts of at most 4,096): their trace_ms is the step launch -- records + in-field witness of call k and chains of call k+1 -- and chain_ms the one chain kernel that starts the train)"
  python tools/sweep.py CONFIG C3-shard-8192-as-8-calls --chunks 8 --steps 6 --warmup 2
  python tools/sweep.py CONFIG C3-shard-8192-as-4-calls-20-steps --batch 2048 --chunks 4 --steps 20 --warmup 5
  python tools/sweep.py CONFIG C3-shard-8192-as-2-calls-20-steps --batch 4096 --chunks 2 --steps 20 --warmup 5
  python tools/sweep.py CONFIG C3-shard-8192-one-call --batch 8192 --steps 6 --warmup 2
  python tools/sweep.py CONFIG C3-shard-8192-one-call-20-steps --batch 8192 --steps 20 --warmup 5
  python tools/sweep.py CONFIG rsa2048-batch-4096 --batch 4096 --steps 20 --warmup 4
  python tools/sweep.py CONFIG C3-shard-8192-serial --batch 8192 --steps 6 --warmup 2 --no-pipeline
  python tools/sweep.py CONFIG C4-rsa4096-w32-4096 --workload rsa4096_w32_e65537 --batch 4096 --steps 4 --warmup 1
  python tools/sweep.py CONFIG C4-rsa4096-w32-4096-serial --workload rsa4096_w32_e65537 --batch 4096 --steps 4 --warmup 1 --no-pipeline
  python tools/sweep.py CONFIG C5-e2048bit-256 --workload rsa2048_e2048bit --batch 256 --steps 8 --warmup 2
  python tools/sweep.py CONFIG C5-e2048bit-256-serial --workload rsa2048_e2048bit --batch 256 --steps 4 --warmup 1 --no-pipeline
  python tools/sweep.py CONFIG rsa1024 --workload rsa1024_e65537 --steps 40 --warmup 4
  python tools/sweep.py CONFIG rsa3072 --workload rsa3072_e65537 --steps 20 --warmup 3
  python tools/sweep.py CONFIG rsa2048-shared-modulus --steps 40 --warmup 4 --shared-modulus
  # BASELINE config 1's shape (ONE signature per call): the latency of one witness, stream-ordered export
  python tools/sweep.py CONFIG C1-one-signature-per-call --batch 1 --steps 200 --warmup 20 --no-pipeline
  python tools/sweep.py CONFIG batch-64-per-call --batch 64 --steps 200 --warmup 20
} > $O/other_configs.txt 2>&1
{
  for wl in rsa2048 rsa3072 rsa4096w32; do python tools/emit_timing.py 1024 $wl 2>&1 | grep kernel; done
  python tools/emit_timing.py 1024 rsa2048 1 noadvice 2>&1 | grep kernel
} > $O/emit_timing.txt
python tools/offpath_timing.py > $O/offpath_kernels.txt 2>&1
python tools/var_exponent_timing.py > $O/var_exponent.txt 2>&1
python tools/sweep.py CONFIG rsa4096-w64 --workload rsa4096_e65537 --steps 20 --warmup 3 >> $O/other_configs.txt 2>&1
python tools/lookup_timing.py > $O/lookup_timing.txt 2>&1
# the caller side (RSASignatureVerifier from message bytes) and the row-program images: kernel stats of the whole verifier, their timing tools
(cd /tmp; timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_verify_msgs -o r -- python $R/bench.py --verify --messages 128 --steps 20 --warmup 3 --no-cpu-baseline --pmc-traffic off > $O/kt_verify_msgs.log 2>&1)
timeout 300 python bench.py --steps 40 --warmup 4 --verify --messages 128 --no-cpu-baseline --pmc-traffic off > $O/bench_verify_from_messages.json 2>/dev/null
python tools/sha256_timing.py > $O/sha256_timing.txt 2>&1
python tools/fresh_advice_timing.py > $O/fresh_advice_timing.txt 2>&1
# [round 4] the advice image as the product: kernel stats under rocprofv3, the bench line, the cells kernel alone by shape
(cd /tmp; timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_advice -o r -- python $R/bench.py --advice --steps 10 --warmup 2 --no-cpu-baseline --pmc-traffic off > $O/kt_advice.log 2>&1)
timeout 600 python bench.py --advice --steps 20 --warmup 5 > $O/bench_advice.json 2> $O/bench_advice.err
if [ -x tools/_bin/cells_bench ]; then
  { LD_LIBRARY_PATH=$R/halo2_rsa_amd/lib timeout 300 tools/_bin/cells_bench 64 2048 1024; LD_LIBRARY_PATH=$R/halo2_rsa_amd/lib timeout 300 tools/_bin/cells_bench 32 4096 431; } > $O/cells_kernel.txt 2>&1
fi
tail -1 $O/bench_pipeline.json | cut -c1-700; tail -1 $O/bench_serial.json | cut -c1-200; tail -1 $O/bench_pipeline_d3s2.json | cut -c1-200; tail -1 $O/bench_torchrun1_config3_shard.json | cut -c1-300; tail -3 $O/bench_torchrun1.err; cat $O/other_configs.txt; cat $O/emit_timing.txt; cat $O/pmc_traffic.json | head -40
