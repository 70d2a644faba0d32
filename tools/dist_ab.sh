#!/bin/bash
# torchrun/RCCL path vs plain path: HW-queue sharing between the caller's stream and the pipeline's side streams
# (H2R_PIPE_STREAM_PRIO = low (default) | normal | high)
TR="python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29611 bench.py --gpus 1 --steps 40 --warmup 4 --no-cpu-baseline"
show() { python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('$1', d['ms_per_step'], d['value'], d['roofline']['avg_launch_ms'], d['roofline']['chain_kernel_avg_ms'])"; }
for prio in low normal high; do
  export H2R_PIPE_STREAM_PRIO=$prio
  H2R_FORCE_DIST=1 $TR 2>/dev/null | grep '^{' | show torchrun-$prio-d2s1
  H2R_FORCE_DIST=1 $TR --pipeline-depth 3 --side-streams 2 2>/dev/null | grep '^{' | show torchrun-$prio-d3s2
  python bench.py --steps 40 --warmup 4 --no-cpu-baseline | show plain-$prio-d2s1
  python bench.py --steps 40 --warmup 4 --no-cpu-baseline --pipeline-depth 3 --side-streams 2 | show plain-$prio-d3s2
done
