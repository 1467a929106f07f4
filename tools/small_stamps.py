#!/usr/bin/env python3
"""Experiment (GPU box): phase boundaries of leaf_fft_small_kernel from s_memtime stamps (a build with -DLEAF_SMALL_STAMP=1:
wave 0 of workgroup (0, 0) overwrites out[0..] with the tick counts since kernel entry; s_memtime ticks at 100 MHz on gfx950).
    python tools/compare_builds.py --build-only stamp:-DLEAF_SMALL_STAMP=1     # in the build container
    python tools/small_stamps.py                                                # on the GPU box"""
import ctypes, os, sys
import torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from leaf_pytorch_amd import _native  # noqa: E402
from leaf_pytorch_amd.initializers import GaborInit  # noqa: E402
so = _native.build(variant="stamp", extra_flags="-DLEAF_SMALL_STAMP=1")
lib = ctypes.CDLL(so); lib.leaf_workspace_bytes.restype = ctypes.c_size_t
dev = torch.device("cuda:0")
F, K, hop, T = 40, 401, 160, 16000
kern = GaborInit(default_window_len=K, sample_rate=16000, min_freq=60.0, max_freq=7800.0)((F, 2)).to(dev)
pw = torch.full((F,), 0.4, device=dev); pb = torch.ones(F, device=dev)
al = torch.full((F,), 0.96, device=dev); de = torch.full((F,), 2.0, device=dev)
ro = torch.full((F,), 2.0, device=dev); ew = torch.full((F,), 0.04, device=dev)
P = lambda t: ctypes.c_void_p(t.data_ptr())
for B in (1, 4):
    x = torch.randn(B, T, device=dev)
    out = torch.empty(B, F, 100, device=dev)
    ws = torch.empty(1 << 16, dtype=torch.uint8, device=dev)
    for it in range(5):
        rc = lib.leaf_forward_f32(P(x), B, T, P(kern), P(pw), P(pb), P(al), P(de), P(ro), P(ew), F, K, hop, 1, 5, P(out), P(ws),
                                  ctypes.c_size_t(ws.numel()), None)
        assert rc == 0, rc
        torch.cuda.synchronize()
    t = out.flatten()[:6].cpu().tolist()
    t2 = out.flatten()[8:14].cpu().tolist()
    if B * F * 2 <= 256:
        print(f"B={B}: second half (split kernel), cycles since ITS entry {t2}")
    names = ["phase0 (twiddles, taps)", "phase1 (forward + table transforms)", "phase2 (filter tasks)", "phase3 (finalize)"]
    prev = 0.0
    # the stamps are SHADER-CLOCK cycles (s_memtime on gfx950 runs with the shader clock: ~2.1 GHz here -- 44 k cycles for a 21 us
    # launch), not the 100 MHz reference clock: convert with the launch's own duration
    print(f"B={B}: cycles since entry {t}")
    for n, v in zip(names, t):
        print(f"   {n:40s} {(v - prev) / 1e3:7.2f} k cycles   (cumulative {v / 1e3:7.2f} k)")
        prev = v
