# same-box A/B of a library variant (halo2_rsa_amd/lib/variants/$1.so, built with python -m halo2_rsa_amd._build $1 <flags>) against the shipped build:
# BASELINE config 2 with the driver's arguments, RSA-1024, config 5 -- assigns/s, ms per step, span / launches, roofline.frac, whole path
V=$PWD/halo2_rsa_amd/lib/variants/$1.so
line() { python -c "
import sys,json
ls=[l for l in sys.stdin if l.startswith('{')]
if not ls: print('$1 FAILED'); sys.exit()
d=json.loads(ls[-1]); print('$1', d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'], d['roofline']['frac'], d.get('whole_path_hbm_frac'))"; }
for i in 1 2 3; do
  for lib in shipped $1; do
    if [ $lib = shipped ]; then unset H2R_LIB; else export H2R_LIB=$V; fi
    python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --pmc-traffic off --scale-anchor off 2>&1 | line "C2 $lib"
  done
done
for i in 1 2; do
  for lib in shipped $1; do
    if [ $lib = shipped ]; then unset H2R_LIB; else export H2R_LIB=$V; fi
    python bench.py --workload rsa1024_e65537 --steps 40 --warmup 5 --no-cpu-baseline --pmc-traffic off --scale-anchor off 2>&1 | line "rsa1024 $lib"
    python bench.py --workload rsa2048_e2048bit --batch 256 --steps 8 --warmup 2 --no-cpu-baseline --pmc-traffic off --scale-anchor off 2>&1 | line "C5 $lib"
  done
done
