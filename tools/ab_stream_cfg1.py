#!/usr/bin/env python3
"""Same-box A/B of the finalize site at BASELINE configs[1] (one clip per workgroup): the tail finalize from LDS sums (what AUTO runs)
against the streaming finalize (LEAF_ALGO_STREAM_FINALIZE), event-timed module calls, interleaved."""
import statistics, sys, os
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from leaf_pytorch_amd import Leaf, _native
m = Leaf().eval().to("cuda:0")
x = 2 * torch.rand(256, 1, 16000, device="cuda:0") - 1
res = {}
variants = {"tail (AUTO)": _native.ALGO_FFT_WG, "stream": _native.ALGO_FFT_WG | _native.ALGO_STREAM_FINALIZE}
with torch.no_grad():
    for rnd in range(6):
        for name, algo in variants.items():
            m._algo = algo
            for _ in range(20):
                m(x)
            torch.cuda.synchronize()
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            for _ in range(200):
                m(x)
            e.record(); e.synchronize()
            if rnd:
                res.setdefault(name, []).append(s.elapsed_time(e) / 200)
    m._algo = variants["tail (AUTO)"]; a = m(x)
    m._algo = variants["stream"]; b = m(x)
for name, v in res.items():
    print(f"{name:14s} {statistics.median(v):.4f} ms per call (min {min(v):.4f})")
print("bit-equal:", bool(torch.equal(a, b)))
