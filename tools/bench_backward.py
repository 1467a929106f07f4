#!/usr/bin/env python3
"""Time forward and forward+backward of the default Leaf (parameters require grad) on one GPU."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from leaf_pytorch_amd import Leaf  # noqa: E402

dev = torch.device("cuda:0")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
torch.manual_seed(0)
m = Leaf().to(dev)
x = 2 * torch.rand(B, 1, 16000, device=dev) - 1


def timed(fn, n=5):
    fn(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n):
        fn()
    e.record(); e.synchronize()
    return s.elapsed_time(e) / n


def fwd():
    with torch.no_grad():
        m(x)


def fwd_bwd():
    m.zero_grad(set_to_none=True)
    m(x).sum().backward()


print(f"B={B}: forward {timed(fwd):.3f} ms   forward+backward {timed(fwd_bwd, 3):.3f} ms")
