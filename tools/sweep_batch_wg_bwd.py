#!/usr/bin/env python3
"""Forward + backward over batch sizes with the workgroup backward kernels forced on (LEAF_WG_BWD_MIN_BLOCKS=0) and off
(a huge threshold): the data behind the backward's batch threshold.  Runs itself twice as a subprocess (the library reads the
switch once).   usage: sweep_batch_wg_bwd.py [sample_rate]"""
import os
import subprocess
import sys
import time

if os.environ.get("LEAF_SWEEP_CHILD"):
    import torch
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from leaf_pytorch_amd import Leaf
    dev = torch.device("cuda:0")
    SR = int(sys.argv[1])
    m = Leaf(sample_rate=SR).to(dev)
    for B in (2, 4, 8, 12, 16, 24, 32, 48, 64, 96, 128):
        x = 2 * torch.rand(B, 1, SR, device=dev) - 1

        def step():
            m.zero_grad(set_to_none=True)
            m(x).sum().backward()
        t0 = time.perf_counter()
        while time.perf_counter() - t0 < 0.2:
            step()
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(20):
            step()
        e.record(); e.synchronize()
        print(B, s.elapsed_time(e) / 20)
    sys.exit(0)

SR = sys.argv[1] if len(sys.argv) > 1 else "16000"
res = {}
for name, thr in (("workgroup", "0"), ("per-wave", "1000000000")):
    out = subprocess.run([sys.executable, os.path.abspath(__file__), SR], env=dict(os.environ, LEAF_SWEEP_CHILD="1", LEAF_WG_BWD_MIN_BLOCKS=thr),
                         capture_output=True, text=True).stdout
    res[name] = {int(l.split()[0]): float(l.split()[1]) for l in out.splitlines() if l and l[0].isdigit()}
for B in sorted(res["workgroup"]):
    w, p = res["workgroup"][B], res["per-wave"].get(B, float("nan"))
    print(f"B={B:4d} forward+backward: workgroup backward {w * 1e3:8.1f} us   per-wave backward {p * 1e3:8.1f} us   best={'wg' if w < p else 'per-wave'}")
