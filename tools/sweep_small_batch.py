#!/usr/bin/env python3
"""Latency of the default forward (40 filters, 16 kHz, 1 s clips) for small batches, FFT vs MFMA algorithm (GPU box):
median of 50 calls each, whole `leaf_forward` (prep + main + finalize).  Feeds the AUTO policy in leaf_kernels.hip."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from leaf_pytorch_amd import Leaf, _native  # noqa: E402

dev = torch.device("cuda:0")
m = Leaf().eval().to(dev)
for p in m.parameters():
    p.requires_grad_(False)
secs = float(sys.argv[1]) if len(sys.argv) > 1 else 1.0
for B in (1, 2, 4, 8, 16, 32, 64, 128, 256):
    x = 2 * torch.rand(B, 1, int(16000 * secs), device=dev) - 1
    row = [f"B={B:4d}"]
    for name, algo in (("fft", _native.ALGO_FFT), ("mfma", _native.ALGO_MFMA)):
        m._algo = algo
        with torch.no_grad():
            for _ in range(10):
                m(x)
            torch.cuda.synchronize()
            evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(50)]
            for s, e in evs:
                s.record(); m(x); e.record()
            torch.cuda.synchronize()
        ms = sorted(s.elapsed_time(e) for s, e in evs)
        row.append(f"{name} {ms[25] * 1e3:8.1f} us")
    auto = {_native.ALGO_FFT: "fft", _native.ALGO_MFMA: "mfma", _native.ALGO_STAGED: "staged"}[
        _native.load().leaf_auto_algo(B, x.shape[-1], 40, 401, 160)]
    print("  ".join(row), f"  auto -> {auto}", flush=True)
