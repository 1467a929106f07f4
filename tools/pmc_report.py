#!/usr/bin/env python3
"""Average the counters of every kernel over the rocprofv3 --pmc passes found under a directory.
   python tools/pmc_report.py gpurun_out/r2d [kernel-substring ...]"""
import collections
import csv
import glob
import json
import re
import os
import sys

root = sys.argv[1]
want = sys.argv[2:] or ["leaf_fft"]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
dur = collections.defaultdict(list)
for path in glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(path)):
        name = r["Kernel_Name"]
        if not any(w in name for w in want):
            continue
        m = re.search(r"(leaf_[a-z_0-9]+|[a-z_0-9]+_kernel)\s*(<[^>]*>)?", name)
        short = (m.group(1) + (m.group(2) or "")) if m else name[:60]
        acc[short][r["Counter_Name"]].append(float(r["Counter_Value"]))
        if "End_Timestamp" in r and r.get("Counter_Name") == "GRBM_GUI_ACTIVE":
            dur[short].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
out = {}
for k, cs in acc.items():
    out[k] = {c: round(sum(v) / len(v), 1) for c, v in sorted(cs.items())}
    out[k]["launches_seen"] = max(len(v) for v in cs.values())
    if dur[k]:
        out[k]["avg_duration_us_under_pmc"] = round(sum(dur[k]) / len(dur[k]) / 1e3, 1)
print(json.dumps(out, indent=1))
