#!/usr/bin/env python3
"""Stress of the in-kernel tail finalize (contiguous dealing + wg_tail_finalize in every workgroup forward kernel): random
geometries and batches; the default forward (clips owned by a workgroup finalized in its tail, the row kernel for the clips that
straddle two workgroups) must equal BIT FOR BIT the forward with the tail compiled out of the dispatch (tools switch
LEAF_FIN_FUSED=0: main kernel + row kernel for every clip, second process) -- one canonical finalize arithmetic.

   usage: stress_tail.py [n_cases [seed]]      needs the tools variant: compare_builds.py --build-only cur:-DLEAF_TOOLS=1"""
import os
import subprocess
import sys
import tempfile

import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from leaf_pytorch_amd import _native  # noqa: E402

VARIANT = os.path.join(REPO, "leaf_pytorch_amd", "build", "variants", "cur", "libleaf_hip.so")
N = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1] != "--ref" else 60
SEED = int(sys.argv[2]) if len(sys.argv) > 2 and sys.argv[1] != "--ref" else 0


def cases(n, seed):
    g = torch.Generator().manual_seed(seed)
    ri = lambda lo, hi: int(torch.randint(lo, hi + 1, (1,), generator=g))
    out = []
    for i in range(n):
        kind = i % 5
        if kind == 0:
            K, hop = [(401, 160), (201, 80), (801, 320)][ri(0, 2)]          # static kernels (2048- and 4096-sample plans)
        elif kind == 4:
            K = ri(417, 1024) * 2 + 1                                       # odd 835 .. 2049: 4096-sample plan, run-time geometry
            hop = ri(int(0.35 * K), K // 2)
        else:
            K = ri(224, 1216)                                               # 2048-sample plan, run-time geometry, odd or even
            hop = ri(max(64, int(0.35 * K)), max(65, K // 2))
        F = [1, 3, 8, 24, 40, 64, 80][ri(0, 6)]
        T = ri(max(K, 1500), 40000)
        B = ri(1, 600)
        out.append((F, K, hop, T, B, ri(0, 3), 1000 * seed + i))
    return out


def run(case, dev):
    F, K, hop, T, B, mode, s = case
    g = torch.Generator().manual_seed(s)
    x = (2 * torch.rand(B, T, generator=g) - 1).to(dev)
    kern = torch.stack([0.1 + torch.rand(F, generator=g) * 2.9, 3.0 + torch.rand(F, generator=g) * K / 4], dim=1).to(dev)
    pw, pb = (0.2 + 0.4 * torch.rand(F, generator=g)).to(dev), torch.ones(F, device=dev)
    pc = [torch.full((F,), v, device=dev) for v in (0.96, 2.0, 2.0, 0.04)]
    return _native.leaf_forward(x, kern, pw, pb, *pc, K, hop, pcen=mode != 1, log1p=mode == 1 and s % 2 == 0, algo=_native.ALGO_FFT_WG)


if __name__ == "__main__":
    if not os.path.exists(VARIANT):
        sys.exit(f"{VARIANT} missing: python tools/compare_builds.py --build-only cur:-DLEAF_TOOLS=1")
    _native.LIB_PATH = VARIANT
    dev = torch.device("cuda:0")
    if len(sys.argv) > 1 and sys.argv[1] == "--ref":
        n, seed, path = int(sys.argv[2]), int(sys.argv[3]), sys.argv[4]
        res = []
        for c in cases(n, seed):
            try:
                res.append(run(c, dev).cpu())
            except RuntimeError as e:
                res.append(str(e))
        torch.save(res, path)
        sys.exit(0)
    with tempfile.TemporaryDirectory() as tmp:
        ref_path = os.path.join(tmp, "ref.pt")
        subprocess.run([sys.executable, os.path.abspath(__file__), "--ref", str(N), str(SEED), ref_path], check=True,
                       env=dict(os.environ, LEAF_FIN_FUSED="0"))
        ref = torch.load(ref_path)
    ran = 0
    for c, r in zip(cases(N, SEED), ref):
        try:
            a = run(c, dev).cpu()
        except RuntimeError as e:
            if not isinstance(r, str):
                sys.exit(f"default forward failed where the reference ran: {c}: {e}")
            continue
        if isinstance(r, str):
            sys.exit(f"reference forward failed where the default ran: {c}: {r}")
        ran += 1
        if not torch.equal(torch.nan_to_num(a, nan=12345.0), torch.nan_to_num(r, nan=12345.0)):
            d = (a.double() - r.double()).abs().max()
            sys.exit(f"NOT bit-identical to the row-kernel-only forward (max diff {float(d):.3e}): {c}")
    print(f"stress_tail: {ran} of {N} cases ran (seed {SEED}; the rest have no workgroup plan), all bit-identical to the "
          f"row-kernel-only forward")
