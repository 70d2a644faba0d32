# Recipe of profiles/r06_queue_probe.txt.  Needs the developer variant: python -m halo2_rsa_amd._build devknobs -DH2R_DEV_KNOBS (on the build host), then
# gpurun -- "bash tools/queue_probe_conditions.sh > gpurun_out/probe.log 2>&1"
cd $GRAFT_REPO_ROOT
export H2R_LIB=$PWD/halo2_rsa_amd/lib/variants/devknobs.so H2R_PROBE_DEBUG=1
ARGS="--gpus 1 --batch 2048 --chunks 4 --steps 20 --warmup 5 --no-cpu-baseline --pmc-traffic off --sub-runs off --scale-anchor off"
one() { # name, env...
  name=$1; shift
  out=$(env "$@" python bench.py $ARGS 2>/tmp/err.txt | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('%.0f %s' % (d['value'], d['config']['pipeline_form']['record_form']))")
  echo "$name: $out | $(grep 'queue probe' /tmp/err.txt | tail -1)"
}
tr() { name=$1; shift
  out=$(env H2R_FORCE_DIST=1 "$@" python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port $((29600 + RANDOM % 300)) bench.py $ARGS 2>/tmp/err.txt | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('%.0f %s' % (d['value'], d['config']['pipeline_form']['record_form']))")
  echo "$name: $out | $(grep 'queue probe' /tmp/err.txt | tail -1)"
}
for i in 1 2; do
one "plain q=default" A=1
one "plain q=4" GPU_MAX_HW_QUEUES=4
one "plain q=8" GPU_MAX_HW_QUEUES=8
tr "torchrun q=4" GPU_MAX_HW_QUEUES=4
tr "torchrun q=8" GPU_MAX_HW_QUEUES=8
tr "torchrun q=8 normal prio" GPU_MAX_HW_QUEUES=8 H2R_PIPE_STREAM_PRIO=normal
tr "torchrun q=4 normal prio" GPU_MAX_HW_QUEUES=4 H2R_PIPE_STREAM_PRIO=normal
one "plain q=2" GPU_MAX_HW_QUEUES=2
done
