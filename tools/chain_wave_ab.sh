#!/bin/bash
# RSA-1024 (16 x 64-bit limbs, K = 32 digits): the one-wave chain (h2r_chain_wave.hpp) against the four-wave chain, same box, developer build.
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(cd "$(dirname "$0")/.." && pwd)
export H2R_LIB=$R/halo2_rsa_amd/lib/variants/devknobs.so
cd $R
for B in 1024 2048 4096 8192; do
  for W in 1 0 1 0; do
    H2R_CHAIN_WAVE=$W timeout 200 python bench.py --workload rsa1024_e65537 --batch $B --steps 20 --warmup 5 --no-cpu-baseline --pmc-traffic off --sub-runs off --scale-anchor off 2>/dev/null |
      python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('B=$B chain_wave=$W  %.2f M assigns/s  ms/step %.4f  record frac %.3f  whole path %.3f  chain_ms %s' % (d['value']/1e6, d['ms_per_step'], d['roofline']['frac'], d.get('whole_path_hbm_frac',0), d['roofline'].get('chain_kernel_avg_ms')))"
  done
done
# the chain kernels alone (no records): stream-ordered pow calls without a trace
for W in 1 0; do H2R_CHAIN_WAVE=$W timeout 200 python tools/chain_half_square_probe.py 2>/dev/null | grep -i "1024" | head -3 | sed "s/^/chain_wave=$W /"; done
