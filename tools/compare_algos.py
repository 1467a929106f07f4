#!/usr/bin/env python3
"""Interleaved timing of the forward algorithms (MFMA direct vs FFT overlap-save) at BASELINE configs[1] size."""
import os
import statistics
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from leaf_pytorch_amd import Leaf, _native  # noqa: E402

dev = torch.device("cuda:0")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
secs = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
torch.manual_seed(0)
m = Leaf().eval().to(dev)
sd = {k: v.detach() for k, v in m.state_dict().items()}
prm = (sd["_complex_conv._kernel"], sd["_pooling.weights"], sd["_pooling._bias"], sd["_compression.alpha"],
       sd["_compression.delta"], sd["_compression.root"], sd["_compression.ema._weights"])
x = 2 * torch.rand(B, 1, int(16000 * secs), device=dev) - 1
algos = {"mfma": _native.ALGO_MFMA, "fft": _native.ALGO_FFT}
times = {k: [] for k in algos}
outs = {}
for rnd in range(7):
    for name, algo in algos.items():
        for _ in range(2):
            outs[name] = _native.leaf_forward(x, *prm, 401, 160, algo=algo)
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(5):
            _native.leaf_forward(x, *prm, 401, 160, algo=algo)
        e.record(); e.synchronize()
        times[name].append(s.elapsed_time(e) / 5)
frames = B * outs["mfma"].shape[2]
for name in algos:
    t = statistics.median(times[name])
    print(f"{name:5s} median {t:.4f} ms  min {min(times[name]):.4f} ms -> {frames / t / 1e3:.2f} M frames/s")
d = (outs["fft"] - outs["mfma"]).abs() / outs["mfma"].abs()
print(f"fft vs mfma max rel diff {float(d.max()):.2e}")
