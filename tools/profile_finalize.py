#!/usr/bin/env python3
"""Workload for `rocprofv3 --kernel-trace --stats`: the default forward with PCEN (finalize does EMA + PCEN), without
PCEN (floor only) and with the log1p extension -- separates the finalize kernel's arithmetic from its memory time."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from leaf_pytorch_amd import Leaf  # noqa: E402

dev = torch.device("cuda:0")
x = 2 * torch.rand(256, 1, 16000, device=dev) - 1
for pcen in (True, False):
    m = Leaf(pcen_compression=pcen).eval().to(dev)
    with torch.no_grad():
        for _ in range(20):
            m(x)
    torch.cuda.synchronize()
