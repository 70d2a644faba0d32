#!/usr/bin/env python3
"""gpurun_out/r06_placement/pass*/ (tools/placement_counters.sh) -> a table: per kernel and counter, mean over the FAST and over the SLOW
dispatches (labels: the SEQ line of each pass's own run.log), their ratio, and the durations of the same dispatches.
usage: tools/placement_counters_parse.py gpurun_out/r06_placement > profiles/r06_placement_counters_raw.txt"""
import collections, csv, glob, json, os, re, sys
root = sys.argv[1]
for d in sorted(glob.glob(os.path.join(root, "pass[0-9]*")), key=lambda p: int(re.search(r"pass(\d+)", p).group(1))):
    log = open(os.path.join(d, "run.log"), errors="replace").read()
    m = re.search(r"^SEQ (\{.*\})$", log, re.M)
    print("==", os.path.basename(d), open(os.path.join(d, "counters.txt")).read().strip())
    if not m:
        print("   no SEQ line (run failed):", log[-300:].replace("\n", " | "))
        continue
    seq = json.loads(m.group(1))
    cc = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
    if not cc:
        print("   no counter_collection.csv")
        continue
    rows = list(csv.DictReader(open(cc[0])))
    for kern, labels in seq["seq"].items():
        # dispatches of this kernel in launch order; a dispatch has one row per counter (and per dimension instance, if any)
        by_disp = collections.OrderedDict()
        for r in rows:
            if kern not in r["Kernel_Name"]:
                continue
            by_disp.setdefault(int(r["Dispatch_Id"]), []).append(r)
        ids = sorted(by_disp)[-len(labels):]
        if len(ids) < len(labels):
            print("   %s: only %d dispatches" % (kern, len(ids))); continue
        acc = collections.defaultdict(lambda: {"F": [], "S": []})
        dur = {"F": [], "S": []}
        for lab, i in zip(labels, ids):
            per = collections.defaultdict(float)
            inst = collections.defaultdict(list)
            for r in by_disp[i]:
                per[r["Counter_Name"]] += float(r["Counter_Value"])
                inst[r["Counter_Name"]].append(float(r["Counter_Value"]))
            for k, v in per.items():
                acc[k][lab].append(v)
            for k, v in inst.items():
                if len(v) > 1:
                    acc[k + " [max/mean over %d instances]" % len(v)][lab].append(max(v) / (sum(v) / len(v)) if sum(v) else 0.0)
            r0 = by_disp[i][0]
            if "Start_Timestamp" in r0 and r0.get("Start_Timestamp"):
                dur[lab].append((int(r0["End_Timestamp"]) - int(r0["Start_Timestamp"])) / 1e3)
        info = seq["lookup" if kern.startswith("lookup") else "trace"]
        print("   %s  (chosen on: fast %s slow %s)" % (kern, info["fast"], info["slow"]))
        if dur["F"]:
            print("      %-60s F %12.1f  S %12.1f  S/F %.3f" % ("duration under the profiler (us)", sum(dur["F"]) / len(dur["F"]), sum(dur["S"]) / len(dur["S"]),
                                                                 (sum(dur["S"]) / len(dur["S"])) / (sum(dur["F"]) / len(dur["F"]))))
        for k in sorted(acc):
            f, s = acc[k]["F"], acc[k]["S"]
            mf, msl = sum(f) / len(f), sum(s) / len(s)
            print("      %-60s F %14.4g  S %14.4g  S/F %s" % (k, mf, msl, "%.3f" % (msl / mf) if mf else "-"))
    # kernel trace durations (separate file)
    kt = glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)
    if kt:
        krows = list(csv.DictReader(open(kt[0])))
        for kern, labels in seq["seq"].items():
            ks = [r for r in krows if kern in r["Kernel_Name"]][-len(labels):]
            if len(ks) == len(labels):
                du = {"F": [], "S": []}
                for lab, r in zip(labels, ks):
                    du[lab].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
                print("   %s kernel-trace duration (us): F %.1f  S %.1f  S/F %.3f" % (kern, sum(du["F"]) / len(du["F"]), sum(du["S"]) / len(du["S"]),
                                                                                     (sum(du["S"]) / len(du["S"])) / (sum(du["F"]) / len(du["F"]))))
