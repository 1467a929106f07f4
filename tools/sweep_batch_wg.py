#!/usr/bin/env python3
"""Per-wave vs workgroup overlap-save kernel over batch sizes (the data behind LEAF_ALGO_AUTO's batch threshold).
   usage: sweep_batch_wg.py [sample_rate]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from leaf_pytorch_amd import Leaf, _native
dev = torch.device('cuda:0')
SR = int(sys.argv[1]) if len(sys.argv) > 1 else 16000
m = Leaf(sample_rate=SR).eval().to(dev)
K, hop = m._pooling.kernel_size, m._pooling.strides
lib = _native.load()
def timed(fn, n=30):
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 0.2: fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); e.synchronize()
    return s.elapsed_time(e) / n
for B in (2, 4, 8, 12, 16, 24, 26, 32, 40, 48, 52, 64, 77, 96, 100, 128, 160, 192, 200, 256, 300, 384, 512):
    x = 2 * torch.rand(B, 1, SR, device=dev) - 1
    res = {}
    for name, algo in (('fft', _native.ALGO_FFT), ('wg', _native.ALGO_FFT_WG)):
        m._algo = algo
        with torch.no_grad():
            res[name] = timed(lambda: m(x))
    auto = lib.leaf_auto_algo(B, SR, 40, K, hop)
    print(f"B={B:4d} per-wave {res['fft']*1e3:7.1f} us  workgroup {res['wg']*1e3:7.1f} us  auto={'wg' if auto==4 else 'fft'}  best={'wg' if res['wg']<res['fft'] else 'fft'}")
