#!/usr/bin/env python3
"""Worst per-column gradient errors of a backward test run (tests/helpers.py writes one JSON line per comparison when
LEAF_GRAD_LOG is set):

    LEAF_GRAD_LOG=gpurun_out/r06/grad_log.jsonl python -m pytest tests/test_gpu_backward.py -m gpu -q
    python tools/summarize_grad_log.py gpurun_out/r06/grad_log.jsonl > profiles/r06/backward_column_errors.txt

`of column max` = max |g - r| / max |r_col| (bound 1e-4); `of entry bound` = max |g - r| / (1e-3 |r_f| + 1e-6 max |r_col|) (bound 1)."""
import collections
import json
import sys

rows = [json.loads(l) for l in open(sys.argv[1])]
worst = collections.defaultdict(lambda: [0.0, 0.0, None, None, 0])
for r in rows:
    w = worst[r["column"]]
    w[4] += 1
    if r["rel_to_col_max"] > w[0]:
        w[0], w[2] = r["rel_to_col_max"], r["ctx"]
    if r["entry_bound_used"] is not None and r["entry_bound_used"] > w[1]:
        w[1], w[3] = r["entry_bound_used"], r["ctx"]
print(f"{len(rows)} gradient-column comparisons against fp64 autograd through the oracle (tests/test_gpu_backward.py)")
print(f"{'column':34s} {'n':>5s} {'of column max':>14s} {'of entry bound':>15s}   worst cases")
for c, w in sorted(worst.items()):
    print(f"{c:34s} {w[4]:5d} {w[0]:14.2e} {w[1]:15.3f}   {w[2]} | {w[3]}")
