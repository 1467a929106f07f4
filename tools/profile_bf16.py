#!/usr/bin/env python3
"""Workload for rocprofv3 --kernel-trace --stats: BASELINE configs[4] shape (256 x 10 s clips) with fp32 or bf16 I/O
(argv[1] = fp32 | bf16) -- which kernel pays for the bf16 path."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from leaf_pytorch_amd import Leaf  # noqa: E402

dev = torch.device("cuda:0")
m = Leaf().eval().to(dev)
x = 2 * torch.rand(256, 1, 160000, device=dev) - 1
if len(sys.argv) > 1 and sys.argv[1] == "bf16":
    x = x.to(torch.bfloat16)
with torch.no_grad():
    for _ in range(12):
        m(x)
torch.cuda.synchronize()
