#!/usr/bin/env python3
"""Interleaved A/B of leaf_backward_f32 (parameter gradients, pooled_raw given) between build variants, through the C ABI.
   usage: [LEAF_CMP_SR=.. LEAF_CMP_B=.. LEAF_CMP_F=.. LEAF_CMP_SECS=.. LEAF_CMP_DX=1 (also dL/dx)] compare_builds_bwd.py name1:-DFLAG name2=prebuilt.so ...
   Prints the median time of the whole backward call and the largest relative difference of the gradients to the first variant."""
import ctypes, os, statistics, sys
import torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from leaf_pytorch_amd.initializers import GaborInit  # noqa: E402
from leaf_pytorch_amd import _native  # noqa: E402
SR = int(os.environ.get("LEAF_CMP_SR", "16000"))
B, F = int(os.environ.get("LEAF_CMP_B", "256")), int(os.environ.get("LEAF_CMP_F", "40"))
T, K, hop = int(SR * float(os.environ.get("LEAF_CMP_SECS", "1"))), int(SR * 25.0 // 1000 + 1), int(SR * 10.0 // 1000)
TP = (T - 1) // hop + 1
DX = int(os.environ.get("LEAF_CMP_DX", "0"))
dev = torch.device("cuda:0")
torch.manual_seed(0)
x = 2 * torch.rand(B, T, device=dev) - 1
kern = GaborInit(default_window_len=K, sample_rate=SR, min_freq=60.0, max_freq=min(7800.0, 0.45 * SR))((F, 2)).to(dev)
pw = torch.full((F,), 0.4, device=dev); pb = torch.ones(F, device=dev)
al = torch.full((F,), 0.96, device=dev); de = torch.full((F,), 2.0, device=dev)
ro = torch.full((F,), 2.0, device=dev); ew = torch.full((F,), 0.04, device=dev)
go = torch.randn(B, F, TP, device=dev)
P = lambda t: ctypes.c_void_p(t.data_ptr())
libs = []
for spec in sys.argv[1:]:
    if "=" in spec.split(":")[0]:
        name, _, so = spec.partition("=")
    else:
        name, _, flags = spec.partition(":")
        so = _native.build(variant=name, extra_flags=flags)
    lib = ctypes.CDLL(os.path.abspath(so))
    lib.leaf_workspace_bytes.restype = ctypes.c_size_t
    lib.leaf_backward_workspace_bytes.restype = ctypes.c_size_t
    libs.append((name, lib))
out = torch.empty(B, F, TP, device=dev); raw = torch.empty(B, F, TP, device=dev)
res, grads = {n: [] for n, _ in libs}, {}
for name, lib in libs:
    ws = torch.empty(lib.leaf_workspace_bytes(B, T, F, K, hop, 0), dtype=torch.uint8, device=dev)
    rc = lib.leaf_forward_save_f32(P(x), B, T, P(kern), P(pw), P(pb), P(al), P(de), P(ro), P(ew), F, K, hop, 1, 0, P(out), P(raw), P(ws),
                                   ctypes.c_size_t(ws.numel()), None)
    assert rc == 0, (name, rc)
    g = [torch.empty(F, 2, device=dev)] + [torch.empty(F, device=dev) for _ in range(6)]
    wb = torch.empty(lib.leaf_backward_workspace_bytes(B, T, F, K, hop, 1, DX), dtype=torch.uint8, device=dev)
    gx = torch.empty(B, T, device=dev) if DX else None

    def call(name=name, lib=lib, g=g, wb=wb, gx=gx, raw=raw.clone()):        # bound NOW: each variant calls its own library
        rc = lib.leaf_backward_f32(P(x), B, T, P(kern), P(pw), P(pb), P(al), P(de), P(ro), P(ew), F, K, hop, 1, P(go), P(raw),
                                   *(P(t) for t in g), P(gx) if DX else None, P(wb), ctypes.c_size_t(wb.numel()), None)
        assert rc == 0, (name, rc)
    grads[name] = (g + ([gx] if DX else []), call)                       # (dL/dx joins the comparison)
for rnd in range(7):
    for name, _ in libs:
        g, call = grads[name]
        for _ in range(3):
            call()
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(10):
            call()
        e.record(); e.synchronize()
        if rnd:
            res[name].append(s.elapsed_time(e) / 10)
base = [t.clone() for t in grads[libs[0][0]][0]]
for name, _ in libs:
    g = grads[name][0]
    diff = max(float((a - b).abs().max() / (b.abs().max() + 1e-30)) for a, b in zip(g, base))
    print(f"{name:14s} backward median {statistics.median(res[name]):.4f} ms  min {min(res[name]):.4f}   max rel-to-max grad diff vs {libs[0][0]} {diff:.2e}")
