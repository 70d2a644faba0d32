#!/usr/bin/env python3
"""Run bench.py under a list of environment settings and print the key timings.
usage: tools/sweep.py ENVVAR v1,v2,... [bench args...]"""
import json, os, subprocess, sys
var, vals, rest = sys.argv[1], sys.argv[2].split(","), sys.argv[3:]
for v in vals:
    env = dict(os.environ); env[var] = v
    out = subprocess.run([sys.executable, "bench.py", "--no-cpu-baseline", "--pmc-traffic", "off"] + [a for a in rest if a not in ("--pmc-traffic", "off")], env=env, capture_output=True, text=True)
    try:
        d = json.loads(out.stdout.strip().splitlines()[-1])
        r = d["roofline"]
        print("%s=%-8s step_ms %.4f value %.0f trace_ms %s (%s GB/s) chain_ms %s %s" % (var, v, d["ms_per_step"], d["value"], r["avg_launch_ms"], r["achieved"], r["chain_kernel_avg_ms"], " ".join(rest)), flush=True)
    except Exception as e:
        print(var, v, "FAILED", out.stderr[-300:])
