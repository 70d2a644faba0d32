#!/usr/bin/env python3
"""Per-kernel statistics of the LAST `steps` launches of each kernel in a rocprofv3 --kernel-trace CSV: the bench's timed
region without the arena's measurement launches, the initialisation call and the warm-up steps that the --stats summary of
the whole process includes.  usage: timed_region_stats.py <dir with *_kernel_trace.csv> <steps> [launches per step]"""
import csv, glob, sys, collections
f = sorted(glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True))[0]
steps = int(sys.argv[2]); per = int(sys.argv[3]) if len(sys.argv) > 3 else 1
by = collections.defaultdict(list)
for r in sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"])):
    if "h2r::" in r["Kernel_Name"]:
        by[r["Kernel_Name"].split("(")[0]].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
print("# last %d launches of each h2r kernel (the timed steps); whole-process numbers are in the --stats CSV next to this file" % (steps * per))
print("# kernel, launches in the process, launches counted, average ns, min ns, max ns")
for k, v in sorted(by.items(), key=lambda kv: -sum(kv[1])):
    w = v[-steps * per:]
    print('"%s",%d,%d,%.1f,%d,%d' % (k, len(v), len(w), sum(w) / len(w), min(w), max(w)))
