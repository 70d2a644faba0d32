R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/tl8192; rm -rf $O; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
export H2R_LIB=$R/halo2_rsa_amd/lib/variants/dev.so
H2R_PIPE_SUB_BATCH=8192 timeout 300 rocprofv3 --kernel-trace --output-format csv -d $O/unsplit -o r -- python $R/bench.py --batch 8192 --steps 6 --warmup 2 --no-cpu-baseline > $O/unsplit.log 2>&1
H2R_PIPE_SUB_BATCH=2048 H2R_PIPE_PACE=1 timeout 300 rocprofv3 --kernel-trace --output-format csv -d $O/sub2048 -o r -- python $R/bench.py --batch 8192 --steps 6 --warmup 2 --no-cpu-baseline > $O/sub2048.log 2>&1
H2R_PIPE_SUB_BATCH=2048 timeout 300 rocprofv3 --kernel-trace --output-format csv -d $O/b2048 -o r -- python $R/bench.py --batch 2048 --steps 12 --warmup 2 --no-cpu-baseline > $O/b2048.log 2>&1
cd $R
for v in unsplit sub2048 b2048; do echo "== $v"; python tools/timeline.py $O/$v | tail -14; done
