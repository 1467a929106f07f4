#!/usr/bin/env python3
"""Forward / forward+backward time of the even-window LEAF geometries (22.05 kHz: K = 552, hop = 220; 11.025 kHz: 276 / 110),
B = 256 x 1 s, next to the 16 kHz default."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from leaf_pytorch_amd import Leaf
dev = torch.device("cuda:0")
for sr in (22050, 11025, 16000):
    m = Leaf(sample_rate=sr).to(dev)
    x = 2 * torch.rand(256, 1, sr, device=dev) - 1
    def run(fn, n=20):
        t0 = time.perf_counter()
        while time.perf_counter() - t0 < 0.25:
            fn()
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(n):
            fn()
        e.record(); e.synchronize()
        return s.elapsed_time(e) / n
    def fwd():
        with torch.no_grad():
            m(x)
    def fb():
        m.zero_grad(set_to_none=True)
        m(x).sum().backward()
    print(f"sr={sr} K={m._complex_conv._kernel_size} hop={m._pooling.strides}: forward {run(fwd):.3f} ms  forward+backward {run(fb):.3f} ms")
