#!/bin/bash
# Round 5 additions to the profile set (gpurun_out/prof_r05 -> profiles/r05_*): the advice image in the prover's representation, the lookup
# argument, the default line with its sub-runs.  rocprofv3 kernel traces and counter passes in separate runs (counters only with --kernel-trace).
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/prof_r05; rm -rf $O; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
A="--steps 20 --warmup 3 --no-cpu-baseline --pmc-traffic off"
for mode in "" "--columns --montgomery"; do
  tag=$( [ -z "$mode" ] && echo advice || echo advice_cm )
  timeout -s KILL 150 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_$tag -o r -- python $R/bench.py --advice $mode $A > $O/kt_$tag.log 2>&1
  timeout -s KILL 150 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmc_${tag}_w -o r -- python $R/bench.py --advice $mode --steps 4 --warmup 1 --no-cpu-baseline --pmc-traffic off --placement-candidates 0 > /dev/null 2>&1
  timeout -s KILL 150 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_${tag}_r -o r -- python $R/bench.py --advice $mode --steps 4 --warmup 1 --no-cpu-baseline --pmc-traffic off --placement-candidates 0 > /dev/null 2>&1
done
timeout -s KILL 150 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_lookup -o r -- python $R/bench.py --lookup --steps 8 --warmup 2 --no-cpu-baseline --pmc-traffic off > $O/kt_lookup.log 2>&1
timeout -s KILL 150 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmc_lookup_w -o r -- python $R/bench.py --lookup --steps 4 --warmup 1 --no-cpu-baseline --pmc-traffic off --placement-candidates 0 > /dev/null 2>&1
timeout -s KILL 150 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_lookup_r -o r -- python $R/bench.py --lookup --steps 4 --warmup 1 --no-cpu-baseline --pmc-traffic off --placement-candidates 0 > /dev/null 2>&1
cd $R
python - <<PY > $O/pmc_traffic_r05.json
import csv, glob, json, os
O = "$O"
out = {}
for tag, kernels in (("advice", ["cells_kernel"]), ("advice_cm", ["cells_kernel"]), ("lookup", ["lookup_fill_kernel", "lookup_setup_kernel"])):
    for k in kernels:
        v = {}
        for c, d in (("WRITE_SIZE", "w"), ("FETCH_SIZE", "r")):
            fs = sorted(glob.glob(os.path.join(O, "pmc_%s_%s" % (tag, d), "**", "*counter_collection.csv"), recursive=True))
            acc = []
            for f in fs[:1]:
                for r in csv.DictReader(open(f)):
                    if r.get("Counter_Name") == c and ("h2r::" + k) in r.get("Kernel_Name", ""):
                        acc.append(float(r["Counter_Value"]))
            v[c + "_KB_per_dispatch"] = sum(acc) / len(acc) if acc else None
            v["dispatches_" + d] = len(acc)
        if v["WRITE_SIZE_KB_per_dispatch"] is not None and v["FETCH_SIZE_KB_per_dispatch"] is not None:
            v["hbm_bytes_per_dispatch"] = int(1024 * (v["WRITE_SIZE_KB_per_dispatch"] + 2 * v["FETCH_SIZE_KB_per_dispatch"]))
        out["%s:%s" % (tag, k)] = v
out["_units"] = "rocprofv3 --pmc, separate passes; KB per dispatch (calibration profiles/r04_pmc_calibration.txt); FETCH_SIZE doubled (gfx950 reports half of wide reads)"
print(json.dumps(out, indent=1))
PY
for t in advice advice_cm lookup; do f=$(ls $O/kt_$t/*/*kernel_stats.csv 2>/dev/null | head -1); [ -n "$f" ] && cp $f $O/kernel_stats_$t.csv; done
timeout -s KILL 400 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_default_driver_args.json 2> $O/bench_default.err
timeout -s KILL 200 python bench.py --advice --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_advice.json 2>/dev/null
timeout -s KILL 200 python bench.py --advice --columns --steps 20 --warmup 5 --no-cpu-baseline --pmc-traffic off > $O/bench_advice_columns.json 2>/dev/null
timeout -s KILL 200 python bench.py --advice --montgomery --steps 20 --warmup 5 --no-cpu-baseline --pmc-traffic off > $O/bench_advice_montgomery.json 2>/dev/null
timeout -s KILL 200 python bench.py --advice --columns --montgomery --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_advice_columns_montgomery.json 2>/dev/null
timeout -s KILL 200 python bench.py --lookup --steps 8 --warmup 2 > $O/bench_lookup.json 2>/dev/null
timeout -s KILL 200 python bench.py --lookup --steps 8 --warmup 2 --placement-candidates 0 --pmc-traffic off > $O/bench_lookup_plain_allocations.json 2>/dev/null
rm -rf $O/kt_* $O/pmc_*_w $O/pmc_*_r
ls -la $O
