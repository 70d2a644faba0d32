#!/usr/bin/env python3
"""Summarise a rocprofv3 --kernel-trace CSV as a timeline: per kernel name the busy time, and the idle gaps
between consecutive launches of the dominant kernel.  usage: tools/timeline.py <dir with *_kernel_trace.csv>"""
import csv, glob, sys, collections
f = sorted(glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True))[0]
rows = list(csv.DictReader(open(f)))
ev = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0][:60]) for r in rows))
t0 = ev[0][0]
byname = collections.defaultdict(list)
for s, e, n in ev: byname[n].append((s, e))
for n, l in sorted(byname.items(), key=lambda kv: -sum(e - s for s, e in kv[1])):
    print("%-62s n=%5d avg %.1f us" % (n, len(l), sum(e - s for s, e in l) / len(l) / 1e3))
dom = max(byname.items(), key=lambda kv: sum(e - s for s, e in kv[1]))
step = [kv for kv in byname.items() if "step_kernel" in kv[0]]   # a pipelined RSA-2048 run: the timed steps are the step launches
if step: dom = step[0]                                            # (the record kernel's launches are mostly the arena's measurements)
l = dom[1][-30:]
print("last launches of", dom[0])
for (s0, e0), (s1, e1) in zip(l, l[1:]):
    between = [(n, (s - t0) / 1e3, (e - s) / 1e3) for s, e, n in ev if e0 - 300000 < s < s1 and n != dom[0]]
    print("  dur %.1f us gap-to-next %.1f us period %.1f us; others: %s" % ((e0 - s0) / 1e3, (s1 - e0) / 1e3, (s1 - s0) / 1e3,
          " ".join("%s@%+.0f(%.0f)" % (n[:12], st - (s0 - t0) / 1e3, du) for n, st, du in between)))
