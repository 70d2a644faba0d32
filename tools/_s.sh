cd $GRAFT_REPO_ROOT
export H2R_LIB=$GRAFT_REPO_ROOT/halo2_rsa_amd/lib/variants/dev.so
for rep in 1 2; do
for d in 20000 26000 32000 38000 45000 52000 60000; do
H2R_TRACE_DYN_LDS=$d python tools/sweep.py LDS $d --steps 60 --warmup 6
done
done
