#!/usr/bin/env python3
"""Rewrite the compile-time switches table at the end of NOTES.md from tools/list_switches.py --markdown (tests/test_host_logic.py
checks that it is current)."""
import os
import subprocess
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tab = subprocess.run([sys.executable, os.path.join(REPO, "tools", "list_switches.py"), "--markdown"], capture_output=True, text=True, check=True).stdout
path = os.path.join(REPO, "NOTES.md")
n = open(path).read()
i = n.index("| switch | default | file | what it selects |")
lines = n[i:].split("\n")
k = 0
while k < len(lines) and lines[k].startswith("|"):
    k += 1
open(path, "w").write(n[:i] + tab.strip() + "\n" + "\n".join(lines[k:]))
print("NOTES.md: switches table rewritten,", tab.count("\n") - 2, "switches")
