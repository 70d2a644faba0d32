# same-box A/B of what sits between consecutive record kernels: store cache-policy bits (dirty L2 at kernel end),
# serialized two-stream pipeline (barrier armed while the previous record kernel runs)
V=$GRAFT_REPO_ROOT/halo2_rsa_amd/lib/variants
for rep in 1 2; do
for v in dev asmnt sc1nt sc01nt sc1; do
  H2R_LIB=$V/$v.so python tools/sweep.py H2R_TAG $v --steps 60 --warmup 6
done
H2R_LIB=$V/dev.so python tools/sweep.py H2R_PIPE_SERIALIZE 1 --steps 60 --warmup 6 --side-streams 2
H2R_LIB=$V/dev.so python tools/sweep.py H2R_PIPE_SERIALIZE 0 --steps 60 --warmup 6 --side-streams 2 --pipeline-depth 3
done
