#!/usr/bin/env python3
"""The last `steps` record-kernel launches of a rocprofv3 --kernel-trace CSV, launch by launch: start, duration, period (start to next start),
overlap with the next launch, and what else ran meanwhile -- the form bench.py's N = 1 line reports as roofline.kernel = trace_kernel<64,32>
(two record kernels in flight on two side streams).  Ends with the figures the roofline is recomputed from.
usage: two_queue_profile.py <dir with *_kernel_trace.csv> <steps> <algorithmic bytes per launch> > profiles/r06_kernel_stats_two_queue_timed.csv"""
import csv, glob, sys
f = sorted(glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True))[0]
steps, algo = int(sys.argv[2]), int(sys.argv[3])
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
ev = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("void ", "").replace("h2r::", ""), r.get("Queue_Id", "")) for r in rows if "h2r::" in r["Kernel_Name"]]
rec = [e for e in ev if e[2].startswith("trace_kernel")][-steps:]
t0 = rec[0][0]
print("# the last %d trace_kernel launches (the timed steps) of %s" % (len(rec), f.split("/")[-1]))
print("launch,queue,start_us,duration_us,period_to_next_start_us,overlap_with_next_us,chain_kernel_started_meanwhile_us")
for i, (s, e, n, q) in enumerate(rec):
    nxt = rec[i + 1] if i + 1 < len(rec) else None
    chains = [c for c in ev if c[2].startswith("chain_kernel") and s <= c[0] < (nxt[0] if nxt else e)]
    print("%d,%s,%.1f,%.1f,%s,%s,%s" % (i, q, (s - t0) / 1e3, (e - s) / 1e3, "%.1f" % ((nxt[0] - s) / 1e3) if nxt else "", "%.1f" % (max(0, e - nxt[0]) / 1e3) if nxt else "",
                                 " ".join("%.1f(%.0f)" % ((c[0] - t0) / 1e3, (c[1] - c[0]) / 1e3) for c in chains)))
d = [e - s for s, e, _, _ in rec]
span = max(e for _, e, _, _ in rec) - min(s for s, _, _, _ in rec)
period = span / len(rec)
ov = [max(0, rec[i][1] - rec[i + 1][0]) for i in range(len(rec) - 1)]
print("# launches %d; average duration %.1f us (min %.1f, max %.1f); period = span / launches = %.1f us; average overlap with the next launch %.1f us; launches in flight %.2f; distinct queues %s"
      % (len(rec), sum(d) / len(d) / 1e3, min(d) / 1e3, max(d) / 1e3, period / 1e3, sum(ov) / max(1, len(ov)) / 1e3, (sum(d) / len(d)) / period, sorted(set(q for _, _, _, q in rec))))
print("# roofline: %d algorithmic bytes per launch / %.1f us period = %.1f GB/s = %.3f of the 8 TB/s peak (per launch duration: %.3f)"
      % (algo, period / 1e3, algo / period, algo / period / 8000.0, algo / (sum(d) / len(d)) / 8000.0))
ch = [e for e in ev if e[2].startswith("chain_kernel")][-steps:]
if ch:
    dc = [e - s for s, e, _, _ in ch]
    print("# chain_kernel: last %d launches, average duration %.1f us" % (len(ch), sum(dc) / len(dc) / 1e3))
st = [e for e in ev if e[2].startswith("step_kernel")]
print("# step_kernel launches in the process: %d (0 = every timed step took the two-queue form)" % len(st))
