#!/usr/bin/env python3
"""Launch timeline of the bench step from a rocprofv3 --kernel-trace CSV: per kernel name the duration, and the idle gaps between
consecutive kernels of the steady state (what a fold of two launches into one could win at most).

    rocprofv3 --kernel-trace --output-format csv -d gpurun_out/r06/kt -o kt -- python bench.py --steps 50 --warmup 10 --no-cpu-baseline
    python tools/kernel_gaps.py gpurun_out/r06/kt > profiles/r06/kernel_gaps.txt"""
import csv
import glob
import os
import statistics
import sys

files = glob.glob(os.path.join(sys.argv[1], "**", "*kernel_trace.csv"), recursive=True)
rows = []
for r in csv.DictReader(open(files[0])):
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0][:60]))
rows.sort()
# steady state: the last 2000 launches
rows = rows[-2000:]
dur, gap = {}, {}
for i, (s, e, n) in enumerate(rows):
    dur.setdefault(n, []).append(e - s)
    if i + 1 < len(rows):
        gap.setdefault((n, rows[i + 1][2]), []).append(rows[i + 1][0] - e)
print("kernel durations (ns): median  p10  p90  count")
for n, v in sorted(dur.items(), key=lambda kv: -sum(kv[1])):
    v.sort()
    print(f"  {n:62s} {statistics.median(v):9.0f} {v[len(v) // 10]:7d} {v[9 * len(v) // 10]:7d} {len(v):6d}")
print("gaps between consecutive kernels (ns): median  p10  p90  count")
for (a, b), v in sorted(gap.items(), key=lambda kv: -len(kv[1])):
    v.sort()
    print(f"  {a[:40]:40s} -> {b[:40]:40s} {statistics.median(v):9.0f} {v[len(v) // 10]:7d} {v[9 * len(v) // 10]:7d} {len(v):6d}")
starts = [s for s, e, n in rows if "leaf_fft_wg_kernel" in n]
if len(starts) > 10:
    per = [b - a for a, b in zip(starts, starts[1:])]
    per.sort()
    print(f"period between main-kernel starts (ns): median {statistics.median(per):.0f}  p10 {per[len(per) // 10]}  p90 {per[9 * len(per) // 10]}")
