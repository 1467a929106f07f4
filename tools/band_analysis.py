#!/usr/bin/env python3
"""How many bins of the 2048-point spectrum does each Gabor filter occupy?  (CPU, numpy; DESIGN.md section 8, "what comes next".)
For every filter of a default-initialised Leaf: the smallest band around its centre bin outside which the K-tap filter's
energy fraction is below eps^2 (eps = 1e-4, 1e-5, 1e-6), and the inverse-transform size a band-limited path would need
(>= twice the band: |y|^2 has twice the bandwidth of y), with a crude cost estimate relative to today's 2048 points per filter.
   usage: band_analysis.py [n_filters sample_rate]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from leaf_pytorch_amd.initializers import GaborInit  # noqa: E402

F = int(sys.argv[1]) if len(sys.argv) > 1 else 40
SR = int(sys.argv[2]) if len(sys.argv) > 2 else 16000
K, N = int(SR * 25.0 // 1000 + 1), 2048 if SR <= 24000 else 4096
kern = GaborInit(default_window_len=K, sample_rate=SR, min_freq=60.0, max_freq=7800.0)((F, 2)).numpy().astype(np.float64)
c = np.sqrt(2 * np.log(2)) / np.pi
t = np.arange(-(K // 2), (K + 1) // 2)
cost = {}
print(f"{F} filters, {SR} Hz, K = {K}, N = {N}: filter, sigma, centre bin, band (bins) at eps = 1e-4 / 1e-5 / 1e-6, inverse size at 1e-6")
for f in range(F):
    mu, sg = np.clip(kern[f, 0], 0, np.pi), np.clip(kern[f, 1], 4 * c, K * c)
    h = np.exp(-t ** 2 / (2 * sg ** 2)) * np.exp(1j * mu * t) / (np.sqrt(2 * np.pi) * sg)
    P = np.abs(np.fft.fft(h, N)) ** 2
    k0, tot, band = int(round(mu * N / (2 * np.pi))), P.sum(), []
    for eps in (1e-4, 1e-5, 1e-6):
        w = next((w for w in range(1, N // 2 + 1) if 1 - P[np.arange(k0 - w, k0 + w + 1) % N].sum() / tot < eps ** 2), N // 2)
        band.append(min(N, 2 * w + 1))
    size = next((m for m in (128, 256, 512, 1024) if m >= 2 * band[2] and m < N), N)
    cost[size] = cost.get(size, 0) + 1
    print(f"  {f:3d}  sigma {sg:6.1f}  bin {k0:4d}   {band[0]:5d} {band[1]:5d} {band[2]:5d}   -> {size}")
rel = sum(n * (m * np.log2(m)) / (N * np.log2(N)) for m, n in cost.items())
print(f"inverse sizes at 1e-6: {dict(sorted(cost.items()))};  transform work ~ {rel:.1f} of {F} full-size transforms ({F / rel:.2f} x less)")
