#!/usr/bin/env python3
"""Where the microseconds of a small-batch eager call go (GPU box): the kernel alone (HIP events around back-to-back launches),
the C-ABI call through ctypes, the dispatcher op, and the whole `Leaf.forward` module call -- each as wall time per call over a
long back-to-back loop (launch-rate bound: host cost per call when the device keeps up) and as the latency of ONE call from
launch to completion (synchronised before and after).  One JSON line per batch size and path."""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from leaf_pytorch_amd import Leaf, _native, _ops  # noqa: E402

dev = torch.device("cuda:0")
m = Leaf().eval().to(dev)
for p_ in m.parameters():
    p_.requires_grad_(False)
sd = m.state_dict()
prm = (sd["_complex_conv._kernel"], sd["_pooling.weights"], sd["_pooling._bias"], sd["_compression.alpha"],
       sd["_compression.delta"], sd["_compression.root"], sd["_compression.ema._weights"])
_ops.load()


def loop_us(fn, n=2000):
    for _ in range(200):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e6


def single_us(fn, n=300):
    for _ in range(50):
        fn()
    ts = []
    for _ in range(n):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        fn()
        torch.cuda.synchronize()
        ts.append((time.perf_counter() - t0) * 1e6)
    ts.sort()
    return ts[len(ts) // 2]


def device_us(fn, n=200):
    for _ in range(50):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


with torch.no_grad():
    for B in (1, 2, 4, 6):
        x = torch.randn(B, 1, 16000, device=dev)
        for name, algo in (("one_launch", _native.ALGO_FFT_SMALL), ("three_launches", _native.ALGO_FFT)):
            m._algo = algo
            paths = {
                "module": lambda: m(x),
                "dispatcher_op": lambda: torch.ops.leaf_amd.forward(x, *prm, 401, 160, False, algo),
                "ctypes": lambda: _native.leaf_forward(x, *prm, 401, 160, algo=algo),
            }
            row = {"B": B, "path": name}
            for pn, fn in paths.items():
                row[pn + "_loop_us"] = round(loop_us(fn), 2)
                row[pn + "_single_us"] = round(single_us(fn), 2)
            row["device_us_back_to_back"] = round(device_us(paths["dispatcher_op"]), 2)
            _, st = _native.leaf_forward_profiled(x, *prm, 401, 160, algo=algo)
            row["stage_ms_prep_main_fin"] = [round(v, 4) for v in st]
            print(json.dumps(row), flush=True)
