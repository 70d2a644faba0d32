#!/bin/bash
# Collect the per-round profile set on the GPU box into gpurun_out/prof_final (copied to profiles/ afterwards).
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/prof_final; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
ARGS="--steps 20 --warmup 3 --no-cpu-baseline"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -o r -- python $R/bench.py $ARGS > $O/kt.log 2>&1
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_serial -o r -- python $R/bench.py $ARGS --no-pipeline > $O/kt_serial.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmc_w -o r -- python $R/bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-pipeline > $O/pmc_w.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_r -o r -- python $R/bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-pipeline > $O/pmc_r.log 2>&1
cd $R
timeout 300 python bench.py --steps 40 --warmup 4 > $O/bench_pipeline.json 2> $O/bench_pipeline.err
timeout 300 python bench.py --steps 40 --warmup 4 --no-pipeline --no-cpu-baseline > $O/bench_serial.json 2>/dev/null
H2R_FORCE_DIST=1 timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29611 bench.py --gpus 1 --steps 10 --warmup 2 --no-cpu-baseline > $O/bench_torchrun1.json 2> $O/bench_torchrun1.err
tail -1 $O/bench_pipeline.json | cut -c1-400; tail -1 $O/bench_serial.json | cut -c1-200; tail -1 $O/bench_torchrun1.json | cut -c1-200; tail -3 $O/bench_torchrun1.err
