# chain workgroups per CU in the RSA-2048 step launch (developer build with -DH2R_DEV_KNOBS: halo2_rsa_amd/lib/variants/dev.so); x2 units: 8 = four per CU (shipped)
export H2R_LIB=$PWD/halo2_rsa_amd/lib/variants/dev.so
for i in 1 2; do
for x2 in 8 4 6 3 2; do
H2R_STEP_CHAIN_X2_PER_CU=$x2 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --pmc-traffic off --scale-anchor off 2>&1 | python -c "
import sys,json
ls=[l for l in sys.stdin if l.startswith('{')]
if not ls: print('x2=$x2 FAILED'); sys.exit()
d=json.loads(ls[-1]); print('x2=$x2', d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'], d['roofline']['frac'], d.get('whole_path_hbm_frac'))"
done; done
