#!/usr/bin/env python3
"""Sweep the fused kernel's de-synchronisation delay (LEAF_ALGO_TUNE_DESYNC) at BASELINE configs[1] size and
print ms per forward (torch events around the whole forward, interleaved rounds, median)."""
import os
import statistics
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from leaf_pytorch_amd import Leaf, _native  # noqa: E402

dev = torch.device("cuda:0")
torch.manual_seed(0)
m = Leaf().eval().to(dev)
sd = {k: v.detach() for k, v in m.state_dict().items()}
prm = (sd["_complex_conv._kernel"], sd["_pooling.weights"], sd["_pooling._bias"], sd["_compression.alpha"],
       sd["_compression.delta"], sd["_compression.root"], sd["_compression.ema._weights"])
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
x = 2 * torch.rand(B, 1, 16000, device=dev) - 1
variants = [-1, 0, 1, 2, 3, 4, 6, 8, 12]
times = {v: [] for v in variants}
for rnd in range(7):
    for v in variants:
        algo = _native.ALGO_MFMA | ((v + 1) << 8)
        for _ in range(2):
            _native.leaf_forward(x, *prm, 401, 160, algo=algo)
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(5):
            _native.leaf_forward(x, *prm, 401, 160, algo=algo)
        e.record(); e.synchronize()
        times[v].append(s.elapsed_time(e) / 5)
for v in variants:
    print(f"desync={v:3d}  median {statistics.median(times[v]):.4f} ms  min {min(times[v]):.4f} ms  -> "
          f"{B * 100 / statistics.median(times[v]) / 1e3:.2f} M frames/s")
