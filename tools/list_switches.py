#!/usr/bin/env python3
"""List the compile-time switches of the HIP sources (`#ifndef LEAF_X / #define LEAF_X default  // comment`): name, default,
file and the comment behind the #define.  Every one of them is an A/B or measurement handle passed as -DLEAF_X=.. through
_native.build(variant=..., extra_flags=...) / tools/compare_builds*.py; the product library is built with the defaults.
   usage: list_switches.py [--markdown]"""
import glob
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
rows = []
for path in sorted(glob.glob(os.path.join(ROOT, "leaf_pytorch_amd", "csrc", "*.h*"))):
    lines = open(path).read().split("\n")
    for i, line in enumerate(lines[:-1]):
        m = re.match(r"\s*#ifndef (LEAF_[A-Z0-9_]+)\s*(//.*)?$", line)
        d = re.match(r"\s*#define (LEAF_[A-Z0-9_]+)\s+(.*?)\s*(//\s*(.*))?$", lines[i + 1])
        if m and d and m.group(1) == d.group(1):
            note = d.group(4) or (m.group(2) or "").lstrip("/ ") or ""
            if not note and i > 0 and lines[i - 1].lstrip().startswith("//"):
                note = lines[i - 1].lstrip().lstrip("/ ")
            rows.append((m.group(1), d.group(2), os.path.basename(path), note))
if "--markdown" in sys.argv:
    print("| switch | default | file | what it selects |\n|---|---|---|---|")
    for r in rows:
        print(f"| `{r[0]}` | `{r[1]}` | `{r[2]}` | {r[3]} |")
else:
    for r in rows:
        print(f"{r[0]:28s} {r[1]:44s} {r[2]:26s} {r[3]}")
