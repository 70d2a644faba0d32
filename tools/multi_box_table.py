#!/usr/bin/env python3
"""A table over default bench.py lines (one JSON line per file, each from a fresh box): headline, plain allocations and the sub-runs.
usage: tools/multi_box_table.py gpurun_out/mb_*.json > profiles/r06_multi_box.txt"""
import json, sys

rows = []
for path in sys.argv[1:]:
    try:
        d = json.loads(open(path).readline())
    except Exception as ex:   # (a box that failed: say so, keep going)
        rows.append((path, None, str(ex)[:60])); continue
    pf = d["config"].get("pipeline_form") or {}
    oc = d.get("other_configs", {})
    g = lambda o, *ks: (lambda v: v if v is not None else float("nan"))(__import__("functools").reduce(lambda a, k: (a or {}).get(k) if isinstance(a, dict) else None, ks, o))
    rows.append((path, d, [d["value"] / 1e6, d["roofline"]["frac"], g(d, "whole_path_hbm_frac"), pf.get("record_form", "?"), pf.get("probe_ms", 0), pf.get("probe_span_ms", 0),
                           g(d, "plain_allocations", "frac"), g(d, "scale_anchor", "roofline_frac"), g(d, "advice", "frac"), g(d, "advice_columns_montgomery", "frac"),
                           g(d, "advice_verify_element", "frac"), g(oc, "C4", "frac"), g(oc, "C5", "frac"), g(oc, "rsa1024_2048_per_call", "whole_path_hbm_frac"),
                           g(d, "records_free_flow", "frac"), g(d, "lookup", "whole_call_frac"), g(d, "lookup", "fill_kernel_frac")]))
hdr = ["M assigns/s", "frac", "whole", "form", "probe ms", "span ms", "plain", "cfg-3 shard", "advice", "adv CM", "verify el", "C4", "C5", "RSA-1024", "rec-free", "lookup", "fill"]
print("box  " + "  ".join("%11s" % h for h in hdr))
for i, (path, d, v) in enumerate(rows):
    if d is None:
        print("%-4d failed: %s (%s)" % (i + 1, v, path)); continue
    print("%-4d " % (i + 1) + "  ".join(("%11.4f" % x) if isinstance(x, float) else ("%11s" % x) for x in v))
