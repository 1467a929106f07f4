#!/usr/bin/env python3
"""Workload for rocprofv3 passes over the training step: a few forward+backward passes of Leaf (parameters require grad).
   usage: profile_backward.py [B [n_filters sample_rate seconds]]   (default 256 clips of the default 40 f / 16 kHz / 1 s)"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from leaf_pytorch_amd import Leaf  # noqa: E402

dev = torch.device("cuda:0")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
F = int(sys.argv[2]) if len(sys.argv) > 2 else 40
SR = int(sys.argv[3]) if len(sys.argv) > 3 else 16000
SECS = float(sys.argv[4]) if len(sys.argv) > 4 else 1.0
torch.manual_seed(0)
m = Leaf(n_filters=F, sample_rate=SR).to(dev)
x = 2 * torch.rand(B, 1, int(SR * SECS), device=dev) - 1
N = int(os.environ.get("LEAF_PROFILE_STEPS", "6"))             # (the first steps run cold: LEAF_PROFILE_STEPS=200 for steady-state averages)
for _ in range(N):
    m.zero_grad(set_to_none=True)
    m(x).sum().backward()
torch.cuda.synchronize()
print("done")
