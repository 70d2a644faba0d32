# same-box sweep of how a large pipelined call is walked: sub-batch size x pacing, per shape (developer build with knobs)
cd $GRAFT_REPO_ROOT
export H2R_LIB=$GRAFT_REPO_ROOT/halo2_rsa_amd/lib/variants/dev.so
run() { # tag, sub, pace, args...
  local tag=$1 sub=$2 pace=$3; shift 3
  H2R_PIPE_SUB_BATCH=$sub H2R_PIPE_PACE=$pace python tools/sweep.py SUBPACE "$tag-sub$sub-pace$pace" "$@"
}
for rep in 1 2; do
for sp in "8192 0" "1024 1" "1024 0" "2048 1" "512 1"; do set -- $sp
  run rsa2048-8192 $1 $2 --batch 8192 --steps 6 --warmup 2
done
for sp in "4096 0" "512 0" "512 1" "1024 0" "1024 1" "2048 0"; do set -- $sp
  run rsa3072-4096 $1 $2 --workload rsa3072_e65537 --batch 4096 --steps 6 --warmup 2
done
for sp in "4096 0" "512 0" "1024 0" "1024 1" "2048 0" "2048 1"; do set -- $sp
  run c4-4096 $1 $2 --workload rsa4096_w32_e65537 --batch 4096 --steps 4 --warmup 1
done
for sp in "8192 0" "1024 1" "1024 0" "2048 1" "2048 0" "4096 1"; do set -- $sp
  run rsa1024-8192 $1 $2 --workload rsa1024_e65537 --batch 8192 --steps 8 --warmup 2
done
for sp in "2048 0" "1024 1" "1024 0"; do set -- $sp
  run rsa2048-2048 $1 $2 --batch 2048 --steps 20 --warmup 4
done
for sp in "3072 0" "1024 1"; do set -- $sp
  run rsa2048-3072 $1 $2 --batch 3072 --steps 12 --warmup 3
done
done
