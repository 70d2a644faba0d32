#!/usr/bin/env python3
"""Turn the two rocprofv3 PMC passes of tools/profile_round.sh (WRITE_SIZE, FETCH_SIZE: separate runs, as
MI355X_MICROARCH.md prescribes) into profiles/pmc_traffic.json: HBM bytes per launch of each h2r kernel.
usage: tools/pmc_to_json.py <prof dir with pmc_w/ and pmc_r/> <batch> > profiles/pmc_traffic.json"""
import collections, csv, glob, json, re, sys
root, batch = sys.argv[1], int(sys.argv[2])

def per_kernel(subs, counter):
    # subs[0]: the serial run (chain kernel, record kernel, in-field kernel); subs[1], when present: the pipelined run, from
    # which only the step launch is taken (the kernels both runs have are reported from the serial one)
    out = {}
    for i, sub in enumerate(subs):
        fs = sorted(glob.glob("%s/%s/**/*counter_collection.csv" % (root, sub), recursive=True))
        if not fs:
            continue
        acc = collections.defaultdict(list)
        for r in csv.DictReader(open(fs[0])):
            if r["Counter_Name"] != counter or "h2r::" not in r["Kernel_Name"]:
                continue
            m = re.search(r"h2r::(\w+)<([^>]*)>", r["Kernel_Name"])
            acc["%s<%s>" % (m.group(1), m.group(2).replace(" ", ""))].append(float(r["Counter_Value"]))
        for k, v in acc.items():
            if i == 0 or k.startswith("step_kernel"):
                out[k] = sum(v) / len(v)
    return out

w, rd = per_kernel(["pmc_w", "pmc_step_w"], "WRITE_SIZE"), per_kernel(["pmc_r", "pmc_step_r"], "FETCH_SIZE")
out = {}
for k in sorted(set(w) | set(rd)):
    wk, rk = w.get(k, 0.0), rd.get(k, 0.0)
    # units: KiB (calibrated on 4 GiB fills / reads, tools/pmc_calibration.sh, profiles/r04_pmc_calibration.txt: WRITE_SIZE 1.0028 per KiB of nt stores,
    # 1.0000 of plain stores; FETCH_SIZE 0.5 per KiB read); gfx950 reports half of wide coalesced reads => x2
    out[k] = {"WRITE_SIZE_KB_per_launch": round(wk, 2), "FETCH_SIZE_KB_per_launch": round(rk, 2),
              "hbm_bytes_per_launch": int(round(1024 * (wk + 2 * rk))), "batch": batch}
out["_note"] = ("rocprofv3 --kernel-trace --pmc WRITE_SIZE and --pmc FETCH_SIZE, separate passes, bench.py --steps 5 "
                "--no-pipeline (step_kernel: the same without --no-pipeline; one launch = the records and the in-field witness of "
                "one call plus the chains of the next), batch %d RSA-2048 e=65537. Units KiB (calibrated on 4 GiB fills and reads, profiles/r04_pmc_calibration.txt: WRITE_SIZE "
                "1.0028 per KiB of nt stores, FETCH_SIZE 0.5 per KiB read). FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 reports half of wide "
                "coalesced reads)." % batch)
json.dump(out, sys.stdout, indent=1)
print()
