# The one-launch step form against the two-queue form with the record kernels alternating between TWO side streams (three buffer sets), per shape;
# developer build halo2_rsa_amd/lib/variants/dev.so (-DH2R_DEV_KNOBS: H2R_PIPE_STEP=0 switches the step form off).  assigns/s, ms per step, whole path.
export H2R_LIB=$PWD/halo2_rsa_amd/lib/variants/dev.so
run() { # name, form, args...
  name=$1; form=$2; shift 2
  if [ $form = step ]; then unset H2R_PIPE_STEP; EX=""; else export H2R_PIPE_STEP=0; EX="--pipeline-depth 3 --side-streams 2"; fi
  python bench.py --gpus 1 --no-cpu-baseline --pmc-traffic off --scale-anchor off $EX "$@" 2>&1 | python -c "
import sys,json
ls=[l for l in sys.stdin if l.startswith('{')]
if not ls: print('$name $form FAILED'); sys.exit()
d=json.loads(ls[-1]); r=d['roofline']; print('%-14s %-5s' % ('$name', '$form'), d['value'], d['ms_per_step'], d.get('whole_path_hbm_frac'))"
}
for i in 1 2; do
for form in step twoq; do
  run C3-4x2048 $form --batch 2048 --chunks 4 --steps 20 --warmup 5
  run rsa1024 $form --workload rsa1024_e65537 --steps 40 --warmup 5
  run rsa3072 $form --workload rsa3072_e65537 --steps 20 --warmup 3
  run rsa4096 $form --workload rsa4096_e65537 --steps 20 --warmup 3
  run C4 $form --workload rsa4096_w32_e65537 --batch 4096 --steps 4 --warmup 1
  run C5 $form --workload rsa2048_e2048bit --batch 256 --steps 8 --warmup 2
done; done
