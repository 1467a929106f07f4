#!/usr/bin/env python3
"""Debug view of the band-limited filter tasks on the GPU: the per-filter records band_tables_kernel wrote (class flags, window
starts), and the event times of the forward with and without the band tasks.   usage: debug_band.py [B] [T]"""
import os
import sys

import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from leaf_pytorch_amd import Leaf, _native  # noqa: E402

DEV = "cuda:0"
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
T = int(sys.argv[2]) if len(sys.argv) > 2 else 16000
lib = _native.load()
torch.manual_seed(0)
m = Leaf().eval().to(DEV)
c = m._compression
P = [m._complex_conv._kernel.detach(), m._pooling.weights.detach(), m._pooling._bias.detach(), c.alpha.detach(), c.delta.detach(), c.root.detach(),
     c.ema._weights.detach()]
x = (2 * torch.rand(B, 1, T) - 1).to(DEV)
F, K, hop = 40, 401, 160
for name, algo in (("band", _native.ALGO_FFT_WG), ("full", _native.ALGO_FFT_WG | _native.ALGO_FULL_TRANSFORMS)):
    if B % 256:
        algo |= _native.algo_reserve_cus(256 - B)
    for _ in range(20):
        out, ms = _native.leaf_forward_profiled(x, *P, K, hop, True, algo)
    acc = [0.0, 0.0, 0.0]
    for _ in range(50):
        out, ms = _native.leaf_forward_profiled(x, *P, K, hop, True, algo)
        acc = [a + b for a, b in zip(acc, ms)]
    print(name, "prep / main (+ band tables) / finalize ms:", [round(a / 50, 4) for a in acc])
    if name == "band":
        nbytes = lib.leaf_workspace_bytes(B, T, F, K, hop, algo)
        ws = _native.workspace(nbytes, torch.device(DEV))
        torch.cuda.synchronize()
        GZ = (64 + K + 256 + 3) // 4 * 4
        al = lambda v: (v + 63) // 64 * 64
        off = al(F * 2048 * 2) + al(F * GZ) + al(F)
        rec = ws.view(torch.int32)[off:off + 4 * F].cpu().view(F, 4)
        print("records (flags, kb256, kb512):")
        for f in range(F):
            print(f"  {f:2d}: flags {int(rec[f, 0])} kb {int(rec[f, 1])} {int(rec[f, 2])}")
        n1 = int((rec[:, 0] & 1).sum()); n2 = int(((rec[:, 0] & 3) == 2).sum())
        print(f"class 256: {n1}, class 512 only: {n2}, full: {F - n1 - n2}")
