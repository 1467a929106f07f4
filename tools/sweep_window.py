#!/usr/bin/env python3
"""FFT vs MFMA algorithm over window lengths (GPU box): Leaf(sample_rate=sr) has K = sr*25/1000 + 1, hop = sr/100;
B = 256 clips of 1 s.  Feeds the AUTO policy's window threshold in leaf_kernels.hip."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from leaf_pytorch_amd import Leaf, _native  # noqa: E402

dev = torch.device("cuda:0")
for sr in (4000, 6000, 8000, 10000, 12000, 16000, 22050, 24000, 32000, 40000, 44100, 48000):
    m = Leaf(sample_rate=sr, init_max_freq=min(7800.0, 0.45 * sr)).eval().to(dev)
    for p in m.parameters():
        p.requires_grad_(False)
    K, hop = m._complex_conv._kernel_size, m._pooling.strides
    x = 2 * torch.rand(256, 1, sr, device=dev) - 1
    row = [f"sr={sr:6d} K={K:5d} hop={hop:4d}"]
    for name, algo in (("fft", _native.ALGO_FFT), ("mfma", _native.ALGO_MFMA)):
        m._algo = algo
        try:
            with torch.no_grad():
                for _ in range(5):
                    m(x)
                torch.cuda.synchronize()
                evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(20)]
                for s, e in evs:
                    s.record(); m(x); e.record()
                torch.cuda.synchronize()
            ms = sorted(s.elapsed_time(e) for s, e in evs)
            row.append(f"{name} {ms[10] * 1e3:8.1f} us")
        except RuntimeError as err:
            row.append(f"{name}   n/a ({str(err)[:30]})")
    auto = {_native.ALGO_FFT: "fft", _native.ALGO_MFMA: "mfma", _native.ALGO_STAGED: "staged"}[
        _native.load().leaf_auto_algo(256, sr, 40, K, hop)]
    print("  ".join(row), f"  auto -> {auto}", flush=True)
