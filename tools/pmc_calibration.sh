#!/bin/bash
# rocprofv3 WRITE_SIZE / FETCH_SIZE against known byte counts (tools/pmc_calibration.hip); prints counter value per KiB moved.
cd /tmp && export TMPDIR=/tmp
BIN=$GRAFT_REPO_ROOT/tools/_bin/pmc_calibration
for c in WRITE_SIZE FETCH_SIZE; do
  rm -rf /tmp/pmc_cal_$c
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d /tmp/pmc_cal_$c -o r -- $BIN > /dev/null 2>&1
  python3 - "$c" <<'PY'
import csv, glob, sys
c = sys.argv[1]
f = sorted(glob.glob("/tmp/pmc_cal_%s/**/*counter_collection.csv" % c, recursive=True))[0]
for r in csv.DictReader(open(f)):
    if r["Counter_Name"] == c:
        name = r["Kernel_Name"].split("(")[0]
        v = float(r["Counter_Value"])
        print("%-10s %-12s counter %.1f  = %.4f per KiB of the 4 GiB the kernel moves (x1024 B: %.4f of the bytes)" % (c, name, v, v / (4 * 1024 * 1024), v * 1024 / (4 << 30)))
PY
done
