#!/usr/bin/env python3
"""Correctness of a build variant of libleaf_hip.so (leaf_pytorch_amd/build/variants/<name>/) before it is timed: the forward of
a few shapes against the CPU oracle and against the in-tree library, through the C ABI.
    python tools/compare_builds.py --build-only pk:-DLEAF_WG_PK=1     # build container
    python tools/check_variant.py pk                                    # GPU box"""
import ctypes, os, sys
import torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from leaf_pytorch_amd import _native  # noqa: E402
from oracle import leaf_oracle as lo  # noqa: E402

name = sys.argv[1]
libs = {}
for n, path in (("tree", _native.LIB_PATH), (name, os.path.join(REPO, "leaf_pytorch_amd", "build", "variants", name, "libleaf_hip.so"))):
    l = ctypes.CDLL(path); l.leaf_workspace_bytes.restype = ctypes.c_size_t
    libs[n] = l
dev = torch.device("cuda:0")
P = lambda t: ctypes.c_void_p(t.data_ptr())
gen = torch.Generator().manual_seed(5)
worst = 0.0
for (F, sr, T, B, pcen) in ((40, 16000, 16000, 256, True), (40, 16000, 16001, 300, False), (7, 16000, 4801, 260, True), (64, 16000, 3300, 64, True),
                           (40, 8000, 8000, 256, True), (80, 32000, 9600, 40, True), (40, 16000, 160000, 26, True)):
    geo = lo.geometry(F, sr)
    K, hop = geo.window_size, geo.hop
    params = lo.default_params(geo, pcen)
    params = {k: v * (1 + 0.1 * (2 * torch.rand(v.shape, generator=gen) - 1)) for k, v in params.items()}
    x = 2 * torch.rand(B, 1, T, generator=gen) - 1
    names = ("_complex_conv._kernel", "_pooling.weights", "_pooling._bias", "_compression.alpha", "_compression.delta", "_compression.root",
             "_compression.ema._weights")
    p = [params[k].reshape(-1).contiguous().to(dev) if k in params else None for k in names]
    p[0] = params[names[0]].contiguous().to(dev)
    xd = x[:, 0].contiguous().to(dev)
    TP = (T - 1) // hop + 1
    outs = {}
    for n, lib in libs.items():
        out = torch.empty(B, F, TP, device=dev)
        ws = torch.empty(lib.leaf_workspace_bytes(B, T, F, K, hop, 4), dtype=torch.uint8, device=dev)
        rc = lib.leaf_forward_f32(P(xd), B, T, P(p[0]), P(p[1]), P(p[2]), *(P(q) if q is not None else None for q in p[3:]), F, K, hop,
                                  1 if pcen else 0, 4, P(out), P(ws), ctypes.c_size_t(ws.numel()), None)
        assert rc == 0, (n, rc)
        torch.cuda.synchronize()
        outs[n] = out.cpu()
    idx = [0, B // 2, B - 1]
    ref = lo.leaf_forward(x[idx], params, geo, pcen, torch.float32)
    rel = lambda a, b: float(((a.double() - b.double()).abs() / b.double().abs().clamp_min(1e-30)).max())
    e_or, e_tree = rel(outs[name][idx], ref), rel(outs[name], outs["tree"])
    worst = max(worst, e_or)
    print(f"F={F} sr={sr} T={T} B={B} pcen={pcen}: {name} vs oracle {e_or:.2e}   tree vs oracle {rel(outs['tree'][idx], ref):.2e}   {name} vs tree {e_tree:.2e}")
    assert torch.isfinite(outs[name]).all() and e_or < 2e-5, "variant out of tolerance"
print("ok, worst rel err vs oracle", f"{worst:.2e}")
