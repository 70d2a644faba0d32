# bench.py --advice: the pipelined export (default) against the same thing composed from plain exports on two torch streams
# (H2R_BENCH_ADV=streams: chain stream at the higher priority; noprio: equal priorities; serial: one stream), alternating on one box
for i in 1 2 3; do
for m in "" streams noprio serial; do
H2R_BENCH_ADV=$m python bench.py --advice --steps 20 --warmup 5 --no-cpu-baseline --pmc-traffic off 2>/dev/null | python -c "
import sys,json
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); r=d['roofline']; print('mode=${m:-export}', d['value'], d['ms_per_step'], r['avg_launch_ms'], r['frac'], d.get('whole_path_hbm_frac'), 'kept', d['config']['buffer_placement']['kept_ms'])"
done; done
