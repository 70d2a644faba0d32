R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/repro; rm -rf $O; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
timeout 200 rocprofv3 --kernel-trace --output-format csv -d $O/kt -o r -- python $R/tools/_repro.py prealloc > $O/log.txt 2>&1
cd $R; tail -8 $O/log.txt
python - <<'PY'
import csv, glob
f = sorted(glob.glob("gpurun_out/repro/kt/**/*kernel_trace.csv", recursive=True))[0]
rows = list(csv.DictReader(open(f)))
ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"][:50], r.get("Queue_Id",""), r.get("Stream_Id","")) for r in rows)
t0 = ev[0][0]
for s, e, n, q, st in ev[-60:]:
    print("%12.1f us  dur %12.1f us  q=%s st=%s  %s" % ((s - t0) / 1e3, (e - s) / 1e3, q, st, n))
PY
