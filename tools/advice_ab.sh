#!/bin/bash
# Same-box A/B of bench.py --advice: residency of cells_kernel (library variants built with -DH2R_CELLS_WAVES=n:
# python -m halo2_rsa_amd._build cells6 -DH2R_CELLS_WAVES=6) x how the call's short kernels are issued (H2R_BENCH_ADV).
for rep in 1 2; do
for lib in "" halo2_rsa_amd/lib/variants/cells6.so halo2_rsa_amd/lib/variants/cells0.so; do
for mode in "" noprio serial; do
  H2R_LIB=${lib:+$PWD/$lib} H2R_BENCH_ADV=$mode timeout 300 python bench.py --advice --steps 20 --warmup 5 --no-cpu-baseline --pmc-traffic off 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('lib=%-8s mode=%-7s %.0f assigns/s  %.4f ms/step  cells %.4f ms frac %.4f  chain %.4f  whole path %.4f' % ('${lib##*/}' or 'default4', '$mode' or 'prio', d['value'], d['ms_per_step'], r['avg_launch_ms'], r['frac'], r['chain_kernel_avg_ms'], d['whole_path_hbm_frac']))"
done; done; done
