#!/usr/bin/env python3
"""Experiment: BASELINE configs[2] (80 filters, 32 kHz, 5 s, 128 clips) through the 2048-sample static kernel with band tasks
(LEAF_NO_4K=1 in the environment) against the 4096-sample kernel: time, parity, classes."""
import os, sys, time
import torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from leaf_pytorch_amd import Leaf, _native
from oracle import leaf_oracle as lo
DEV = "cuda:0"
torch.manual_seed(0)
m = Leaf(n_filters=80, sample_rate=32000).eval().to(DEV)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 128
x = (2 * torch.rand(B, 1, 160000) - 1).to(DEV)
cls = _native.band_classes(m._complex_conv._kernel.detach(), m._pooling.weights.detach(), 801, 320)
if cls is not None:
    c = cls.cpu().tolist(); print("classes:", {k: c.count(k) for k in (256, 512, 2048)})
print("auto algo", _native.load().leaf_auto_algo(B, 160000, 80, 801, 320), "plan", _native.fft_plan_info(B, 160000, 80, 801, 320))
for name, algo in (("band", _native.ALGO_FFT_WG), ("full", _native.ALGO_FFT_WG | _native.ALGO_FULL_TRANSFORMS)):
    m._algo = algo
    with torch.no_grad():
        for _ in range(10): out = m(x)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(30): out = m(x)
        torch.cuda.synchronize()
    print(f"{name}: {(time.perf_counter() - t0) / 30 * 1e3:.4f} ms")
    params = {k: v.cpu() for k, v in m.state_dict().items()}
    ref = lo.leaf_forward(x[:1].cpu(), params, lo.geometry(80, 32000), True, torch.float64)
    print(f"   vs oracle (clip 0): {float(((out[:1].cpu().double() - ref).abs() / ref.abs()).max()):.2e}")
