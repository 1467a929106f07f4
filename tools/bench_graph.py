#!/usr/bin/env python3
"""Small-batch latency: eager launches vs one HIP-graph replay of the whole forward."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from leaf_pytorch_amd import Leaf  # noqa: E402

dev = torch.device("cuda:0")
for B in (1, 4, 16, 64, 256):
    torch.manual_seed(0)
    m = Leaf().eval().to(dev)
    for p in m.parameters():
        p.requires_grad_(False)
    x = 2 * torch.rand(B, 1, 16000, device=dev) - 1
    with torch.no_grad():
        for _ in range(3):
            m(x)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            out = m(x)

        def timed(fn, n=50):
            fn(); torch.cuda.synchronize()
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            for _ in range(n):
                fn()
            e.record(); e.synchronize()
            return s.elapsed_time(e) / n * 1e3

        print(f"B={B:4d}: eager {timed(lambda: m(x)):8.1f} us   graph replay {timed(g.replay):8.1f} us")
