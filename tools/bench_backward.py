#!/usr/bin/env python3
"""Time forward and forward+backward of Leaf (parameters require grad) on one GPU.
   usage: bench_backward.py [B [n_filters sample_rate seconds [nodx]]]   (default 256 clips of the default 40 f / 16 kHz / 1 s;
   nodx skips the dL/dx timing -- staged kernels, hundreds of ms, for geometries without a fused dL/dx)"""
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


def timed(fn, n=20):
    import time
    t0 = time.perf_counter()                      # spin-up: let the clocks settle before timing
    while time.perf_counter() - t0 < 0.25:
        fn()
    torch.cuda.synchronize()
    reps = []                                     # median of five event-timed regions (a host hiccup costs one region, not the figure)
    for _ in range(5):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(n):
            fn()
        e.record(); e.synchronize()
        reps.append(s.elapsed_time(e) / n)
    return sorted(reps)[2]


def fwd():
    with torch.no_grad():
        m(x)


def fwd_bwd():
    m.zero_grad(set_to_none=True)
    m(x).sum().backward()


# the same step with the gradient of the output already resident, as it arrives from a backbone in training
# (train.py:257-259): `.sum().backward()` above adds a reduction, a fill and a broadcast copy of its own (~25 us at cfg1)
with torch.no_grad():
    go = torch.randn_like(m(x))


def fwd_bwd_resident():
    m.zero_grad(set_to_none=True)
    torch.autograd.backward(m(x), go)


xg = x.clone().requires_grad_(True)


def fwd_bwd_dx():
    m.zero_grad(set_to_none=True)
    xg.grad = None
    m(xg).sum().backward()


def graphed(fn_step):
    """the step captured into one HIP graph (torch.cuda.graph): what is left of the step when the host's dispatcher / autograd work is
    taken out -- for the small batches where that work is longer than the kernels"""
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(3):
            fn_step()
    torch.cuda.current_stream().wait_stream(side)
    g = torch.cuda.CUDAGraph()
    m.zero_grad(set_to_none=True)
    with torch.cuda.graph(g):
        torch.autograd.backward(m(x), go)
    return g.replay


NODX = len(sys.argv) > 5 and sys.argv[5] == "nodx"
print(f"B={B} F={F} sr={SR} {SECS:g}s: forward {timed(fwd):.3f} ms   forward+backward {timed(fwd_bwd):.3f} ms   "
      f"(grad_out resident: {timed(fwd_bwd_resident):.3f} ms)   "
      + ("" if NODX else f"forward+backward incl. dL/dx {timed(fwd_bwd_dx):.3f} ms")
      + (f"   as one HIP graph (grad_out resident): {timed(graphed(fwd_bwd_resident)):.3f} ms" if os.environ.get("LEAF_BENCH_GRAPH") else ""))
