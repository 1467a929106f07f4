#!/usr/bin/env python3
"""Workload for rocprofv3 passes on ONE algorithm: LEAF_PROFILE_ALGO (default 4 = workgroup FFT kernel) x 10 forwards at
BASELINE configs[1] size (256 x 1 s, default Leaf), after a short spin-up.
   rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY ... --kernel-trace --output-format csv -d out -o w -- python tools/profile_algo.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from leaf_pytorch_amd import Leaf  # noqa: E402

dev = torch.device("cuda:0")
torch.manual_seed(0)
m = Leaf().eval().to(dev)
for p in m.parameters():
    p.requires_grad_(False)
m._algo = int(os.environ.get("LEAF_PROFILE_ALGO", "4"))
B = int(os.environ.get("LEAF_PROFILE_BATCH", "256"))
x = 2 * torch.rand(B, 1, 16000, device=dev) - 1
with torch.no_grad():
    for _ in range(int(os.environ.get("LEAF_PROFILE_ITERS", "10"))):
        m(x)
torch.cuda.synchronize()
print("done")
