export H2R_LIB=$PWD/halo2_rsa_amd/lib/variants/knobs.so
for wl in rsa1024_e65537 rsa3072_e65537 rsa4096_e65537; do for f in 0 1; do
  a=$(H2R_VERIFY_FOLD=$f python bench.py --verify --workload $wl --steps 30 --warmup 4 --no-cpu-baseline --pmc-traffic off 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value']), d['ms_per_step'])")
  b=$(H2R_VERIFY_FOLD=$f python bench.py --verify --messages 128 --workload $wl --steps 30 --warmup 4 --no-cpu-baseline --pmc-traffic off 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value']), d['ms_per_step'])")
  m=$(python bench.py --workload $wl --steps 30 --warmup 4 --no-cpu-baseline --pmc-traffic off 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value']), d['ms_per_step'])")
  echo "$wl fold=$f  digests: $a   messages: $b   modpow: $m"
done; done
