cd $GRAFT_REPO_ROOT
export H2R_LIB=halo2_rsa_amd/lib/variants/knobs.so
W="--workload rsa2048_e2048bit --batch 256 --steps 8 --warmup 2"
python tools/sweep.py H2R_TRACE_DYN_LDS -1,20000,32000,52000,64000,80000 $W
python tools/sweep.py H2R_CHAIN_PRIO 1 $W
python tools/sweep.py H2R_TRACE_PRIO 1 $W
python tools/sweep.py H2R_PIPE_STREAM_PRIO normal,high $W
