#!/usr/bin/env python3
"""Forward and forward+backward time of the default Leaf (40 filters, 25 ms window, 10 ms hop) on 256 x 1 s clips at every
sample rate LEAF front ends are commonly built for: one JSON line per rate (window, hop, algorithm AUTO picks, ms, frames/s).
   usage: bench_rates.py [B]"""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from leaf_pytorch_amd import Leaf, _native  # noqa: E402

dev = torch.device("cuda:0")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
NAMES = {_native.ALGO_FFT: "fft (per-wave)", _native.ALGO_MFMA: "mfma", _native.ALGO_STAGED: "staged",
         _native.ALGO_FFT_WG: "fft_wg (workgroup per block)"}


def timed(fn, n=20):
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 0.25:        # spin-up: let the clocks settle
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n):
        fn()
    e.record(); e.synchronize()
    return s.elapsed_time(e) / n


lib = _native.load()
for sr in (8000, 11025, 16000, 22050, 24000, 32000, 44100, 48000):
    torch.manual_seed(0)
    m = Leaf(n_filters=40, sample_rate=sr).to(dev)
    x = 2 * torch.rand(B, 1, sr, device=dev) - 1
    K, hop = m._pooling.kernel_size, m._pooling.strides

    def fwd():
        with torch.no_grad():
            m(x)

    def fwd_bwd():
        m.zero_grad(set_to_none=True)
        m(x).sum().backward()

    f_ms, fb_ms = timed(fwd), timed(fwd_bwd)
    frames = B * ((sr - 1) // hop + 1)
    print(json.dumps({"sample_rate": sr, "window": K, "hop": hop, "batch": B, "algo_auto": NAMES.get(lib.leaf_auto_algo(B, sr, 40, K, hop)),
                      "forward_ms": round(f_ms, 4), "forward_backward_ms": round(fb_ms, 4),
                      "forward_frames_per_s": round(frames / f_ms * 1e3), "train_frames_per_s": round(frames / fb_ms * 1e3)}))
