#!/usr/bin/env python3
"""Per-GPU timing of the other BASELINE.json configs (they are parity cases, not bench.py lines).  One GPU,
per-GPU batch of each config; prints one JSON line per config."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from leaf_pytorch_amd import Leaf, _native  # noqa: E402

dev = torch.device("cuda:0")
CONFIGS = [
    ("cfg0 default B=4x1s", dict(n_filters=40, sample_rate=16000), 4, 1.0, True),
    ("cfg1 default B=256x1s", dict(n_filters=40, sample_rate=16000), 256, 1.0, True),
    ("cfg2 80f/32k/5s per-GPU B=128", dict(n_filters=80, sample_rate=32000), 128, 5.0, True),
    ("cfg3 PCEN off B=512x1s", dict(n_filters=40, sample_rate=16000), 512, 1.0, False),
    ("cfg4 40f/16k/10s per-GPU B=256 (fp32 I/O)", dict(n_filters=40, sample_rate=16000), 256, 10.0, True),
    ("audioset-cfg 64f/16k/1s B=256", dict(n_filters=64, sample_rate=16000), 256, 1.0, True),
]
for name, kw, B, secs, pcen in CONFIGS:
    torch.manual_seed(0)
    m = Leaf(pcen_compression=pcen, **kw).eval().to(dev)
    for p in m.parameters():
        p.requires_grad_(False)
    T = int(kw["sample_rate"] * secs)
    x = 2 * torch.rand(B, 1, T, device=dev) - 1
    with torch.no_grad():
        for _ in range(3):
            out = m(x)
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        n = 10
        s.record()
        for _ in range(n):
            out = m(x)
        e.record(); e.synchronize()
    ms = s.elapsed_time(e) / n
    frames = out.shape[0] * out.shape[2]
    K, hop, F = m._complex_conv._kernel_size, m._pooling.strides, kw["n_filters"]
    flops = (2 * 2 * F * K * hop + 2 * F * K) * frames
    print(json.dumps({"config": name, "in": list(x.shape), "out": list(out.shape), "ms": round(ms, 3),
                      "frames_per_s": round(frames / ms * 1e3), "algorithmic_TFLOPs": round(flops / ms / 1e9, 1)}), flush=True)
