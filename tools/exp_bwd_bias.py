#!/usr/bin/env python3
"""Round 6: the backward's band tasks with the forward's bias-aware class decision (the default) against round 5's strict decision
(LEAF_FLAG_BWD_STRICT_BAND_CLASSES) and against full transforms: per gradient column `error of the column's largest entry / share of the
per-filter bound used` against fp64 autograd through the oracle, and the time of the backward call (incl. the host-to-device copies of
this script: compare the rows, not the absolute figures)."""
import ctypes, os, sys, time
import torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "tests"))
from leaf_pytorch_amd import _native
from oracle import leaf_oracle as lo
from helpers import grad_errors
DEV = "cuda:0"
names = ["_complex_conv._kernel", "_pooling.weights", "_pooling._bias", "_compression.alpha", "_compression.delta", "_compression.root", "_compression.ema._weights"]
lib = _native.load()

def backward(x, args, K, hop, go, flags, need_dx=False):
    B, T = x.shape[0], x.shape[-1]; F = args[0].shape[0]
    x2 = x.reshape(B, T).contiguous()
    outs = [torch.empty_like(args[0]), torch.empty(F, device=DEV), torch.empty(F, device=DEV)] + [torch.empty(F, device=DEV) for _ in range(4)]
    gx = torch.empty_like(x2) if need_dx else None
    fl = _native.FLAG_PCEN | flags
    ws = torch.empty(max(lib.leaf_backward_workspace_bytes(B, T, F, K, hop, fl, int(need_dx)), 4), dtype=torch.uint8, device=DEV)
    P = lambda t: None if t is None else ctypes.c_void_p(t.data_ptr())
    rc = lib.leaf_backward_f32(P(x2), B, T, *[P(a.reshape(-1).contiguous() if a.dim() != 2 else a) for a in args], F, K, hop, fl, P(go), None,
                               *[P(o) for o in outs], P(gx), P(ws), ws.numel(), _native.stream_ptr(torch.device(DEV)))
    assert rc == 0, rc
    return outs, gx

for sr, F, B, T in ((16000, 40, 36, 16000), (16000, 40, 170, 3300), (32000, 80, 96, 9600)):
    gen = torch.Generator().manual_seed(sr + B)
    geo = lo.geometry(F, sr)
    K, hop = geo.window_size, geo.hop
    params = lo.default_params(geo, True)
    params = {k: (v * (1 + 0.05 * (2 * torch.rand(v.shape, generator=gen) - 1)) if "kernel" not in k else v) for k, v in params.items()}
    x = torch.randn(B, 1, T, generator=gen)
    go = torch.randn(B, F, (T - 1) // hop + 1, generator=gen)
    p64 = {k: v.double().requires_grad_(True) for k, v in params.items()}
    lo.leaf_forward(x.double(), p64, geo, True, torch.float64).backward(go.double())
    args = [params[k].to(DEV) for k in names]
    for tag, flags in (("strict", _native.FLAG_BWD_STRICT_BAND_CLASSES), ("bias-aware", 0), ("full", _native.FLAG_BWD_FULL_TRANSFORMS)):
        outs, _ = backward(x.to(DEV), args, K, hop, go.to(DEV), flags)
        worst = {}
        for n, g in zip(names, outs):
            for label, ea, eb in grad_errors(n, g.cpu().reshape(p64[n].grad.shape), p64[n].grad):
                worst[label.split(".")[-1]] = (ea, eb)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(20):
            backward(x.to(DEV), args, K, hop, go.to(DEV), flags)
        torch.cuda.synchronize(); ms = (time.perf_counter() - t0) / 20 * 1e3
        print(f"{sr} Hz B {B} T {T} {tag:10s}: {ms:.3f} ms  " + "  ".join(f"{k} {a:.1e}/{b:.2f}" for k, (a, b) in worst.items()))
