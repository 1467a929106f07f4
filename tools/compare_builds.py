#!/usr/bin/env python3
"""Time the default forward (B=256 x 1 s) for several builds of csrc (extra hipcc flags per variant), interleaved.
   usage: [LEAF_CMP_ALGO=3|4] compare_builds.py name1:-DFLAG=1 name2:-DFLAG=0 name3=prebuilt.so ...
   (LEAF_CMP_ALGO selects the algorithm: 0 AUTO (default), 3 per-wave FFT kernel, 4 workgroup FFT kernel)
   Variants are built by leaf_pytorch_amd._native.build(variant=name) into leaf_pytorch_amd/build/variants/<name>/ (they travel
   with gpurun, so `compare_builds.py --build-only name:flags ...` in the build container saves GPU-box time)."""
import ctypes, os, statistics, subprocess, sys
import torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from leaf_pytorch_amd.initializers import GaborInit  # noqa: E402
from leaf_pytorch_amd import _native  # noqa: E402
argv = [a for a in sys.argv[1:] if a != "--build-only"]
if "--build-only" in sys.argv:
    for spec in argv:
        name, _, flags = spec.partition(":")
        print(_native.build(variant=name, extra_flags=flags))
    sys.exit(0)
ALGO = int(os.environ.get("LEAF_CMP_ALGO", "0"))
SR = int(os.environ.get("LEAF_CMP_SR", "16000"))                  # LEAF_CMP_SR: the default front end at another sample rate
# LEAF_CMP_B / LEAF_CMP_F / LEAF_CMP_SECS: batch, filters, clip length (BASELINE configs[2]: LEAF_CMP_SR=32000 LEAF_CMP_B=128 LEAF_CMP_F=80 LEAF_CMP_SECS=5)
B, F = int(os.environ.get("LEAF_CMP_B", "256")), int(os.environ.get("LEAF_CMP_F", "40"))
T, K, hop = int(SR * float(os.environ.get("LEAF_CMP_SECS", "1"))), int(SR * 25.0 // 1000 + 1), int(SR * 10.0 // 1000)
dev = torch.device("cuda:0")
torch.manual_seed(0)
x = 2 * torch.rand(B, T, device=dev) - 1
kern = GaborInit(default_window_len=K, sample_rate=SR, min_freq=60.0, max_freq=min(7800.0, 0.45 * SR))((F, 2)).to(dev)
pw = torch.full((F,), 0.4, device=dev); pb = torch.ones(F, device=dev)
al = torch.full((F,), 0.96, device=dev); de = torch.full((F,), 2.0, device=dev)
ro = torch.full((F,), 2.0, device=dev); ew = torch.full((F,), 0.04, device=dev)
out = torch.empty(B, F, (T + hop - 1) // hop, device=dev)
P = lambda t: ctypes.c_void_p(t.data_ptr())
libs = []
for spec in argv:
    if "=" in spec.split(":")[0]:                       # prebuilt library
        name, _, so = spec.partition("=")
        lib = ctypes.CDLL(os.path.abspath(so)); lib.leaf_workspace_bytes.restype = ctypes.c_size_t
        libs.append((name, lib))
        continue
    name, _, flags = spec.partition(":")
    so = _native.build(variant=name, extra_flags=flags)
    lib = ctypes.CDLL(so); lib.leaf_workspace_bytes.restype = ctypes.c_size_t
    libs.append((name, lib))
ws = torch.empty(max(l.leaf_workspace_bytes(B, T, F, K, hop, 0) for _, l in libs), dtype=torch.uint8, device=dev)
ms = (ctypes.c_float * 3)()
res = {n: [] for n, _ in libs}
prep = {n: [] for n, _ in libs}
for rnd in range(7):
    for name, lib in libs:
        for _ in range(4):
            rc = lib.leaf_forward_profiled_f32(P(x), B, T, P(kern), P(pw), P(pb), P(al), P(de), P(ro), P(ew), F, K, hop, 1, ALGO,
                                               P(out), P(ws), ctypes.c_size_t(ws.numel()), None, ms)
            assert rc == 0, (name, rc)
            if rnd:
                res[name].append(ms[1])
                prep[name].append(ms[0])
# whole forward, back to back on the default stream (includes launch gaps between its kernels)
whole = {n: [] for n, _ in libs}
for rnd in range(5):
    for name, lib in libs:
        for _ in range(3):
            lib.leaf_forward_f32(P(x), B, T, P(kern), P(pw), P(pb), P(al), P(de), P(ro), P(ew), F, K, hop, 1, ALGO, P(out), P(ws),
                                 ctypes.c_size_t(ws.numel()), None)
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(20):
            lib.leaf_forward_f32(P(x), B, T, P(kern), P(pw), P(pb), P(al), P(de), P(ro), P(ew), F, K, hop, 1, ALGO, P(out), P(ws),
                                 ctypes.c_size_t(ws.numel()), None)
        e.record(); e.synchronize()
        whole[name].append(s.elapsed_time(e) / 20)
for name, _ in libs:
    print(f"{name:20s} table launch median {statistics.median(prep[name]):.4f} ms   main kernel median {statistics.median(res[name]):.4f} ms  min {min(res[name]):.4f}   "
          f"whole forward median {statistics.median(whole[name]):.4f} ms")
