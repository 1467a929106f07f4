#!/usr/bin/env python3
"""Workload for PMC passes on the FFT path: a few default-algo forwards at BASELINE configs[1] size."""
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
x = 2 * torch.rand(256, 1, 16000, device=dev) - 1
with torch.no_grad():
    for _ in range(5):
        m(x)
torch.cuda.synchronize()
