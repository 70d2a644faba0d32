#!/usr/bin/env python3
"""Per-kernel statistics of the LAST `steps` launches of each kernel in a rocprofv3 --kernel-trace CSV: the bench's timed
region without the arena's measurement launches, the initialisation call and the warm-up steps that the --stats summary of
the whole process includes.  usage: timed_region_stats.py <dir with *_kernel_trace.csv> <steps> [launches per step]
Launches of one kernel may OVERLAP (record kernels alternating between two side streams): next to a launch's own duration the
table gives the launches' PERIOD -- (end of the last counted launch - start of the first) / launches -- which is what bench.py's
roofline takes for such a kernel."""
import csv, glob, sys, collections
f = sorted(glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True))[0]
steps = int(sys.argv[2]); per = int(sys.argv[3]) if len(sys.argv) > 3 else 1
by = collections.defaultdict(list)
for r in sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"])):
    if "h2r::" in r["Kernel_Name"]:
        by[r["Kernel_Name"].split("(")[0]].append((int(r["Start_Timestamp"]), int(r["End_Timestamp"])))
print("# last %d launches of each h2r kernel (the timed steps); whole-process numbers are in the --stats CSV next to this file" % (steps * per))
print("# kernel, launches in the process, launches counted, average duration ns, min ns, max ns, period ns (span of the counted launches / launches), launches in flight (duration / period)")
for k, v in sorted(by.items(), key=lambda kv: -sum(e - s for s, e in kv[1])):
    w = v[-steps * per:]
    d = [e - s for s, e in w]
    period = (max(e for s, e in w) - min(s for s, e in w)) / len(w)
    print('"%s",%d,%d,%.1f,%d,%d,%.1f,%.2f' % (k, len(v), len(w), sum(d) / len(d), min(d), max(d), period, (sum(d) / len(d)) / period if period else 0))
