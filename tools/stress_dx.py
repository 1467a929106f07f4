#!/usr/bin/env python3
"""Stress of the dL/dx kernels on the workgroup structure (G shared in LDS, filters added in ticket order): random geometries and
batches large enough for them, each backward run three times -- the three results must be bit-identical (the sum order does
not depend on timing) -- and compared with the kernels that run with the workgroup dL/dx switched off: round 2's block-per-wave
kernel at the static geometries, the staged path at run-time geometries since round 4 (a second process with the tools
switches LEAF_WGG_BWD_DX=0 LEAF_WG_BWD_DX=0), which share no accumulation code with them.

With LEAF_STRESS=bwd4k: the static 32 kHz backward on 4096-sample blocks (parameter gradients) against the static
2048-sample kernel (LEAF_4K_BWD_STATIC=0 in the second process).

   usage: [LEAF_STRESS=bwd4k] stress_dx.py [n_cases [seed]]     needs the tools variant: compare_builds.py --build-only cur:-DLEAF_TOOLS=1
   Every backward is a few ms; run the whole thing under `timeout` (a hang would be a deadlock in the ticket protocol)."""
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
BWD4K = os.environ.get("LEAF_STRESS", "") == "bwd4k"


def cases(n, seed):
    g = torch.Generator().manual_seed(seed)
    ri = lambda lo, hi: int(torch.randint(lo, hi + 1, (1,), generator=g))
    out = []
    for i in range(n):
        if BWD4K:
            F = [1, 2, 5, 8, 24, 40, 80][ri(0, 6)]
            T = ri(801, 40000)
            B = min(600, max(2, -(-ri(130, 500) // -(-T // 3200))))
            out.append((F, 801, 320, T, B, ri(0, 1) == 1, 1000 * seed + i))
            continue
        if i % 4 == 0:
            K, hop = [(401, 160), (201, 80)][ri(0, 1)]
        else:
            K = ri(224, 1216)
            hop = ri(max(64, int(0.35 * K)), max(65, K // 2))
        F = [1, 2, 3, 5, 8, 13, 24, 40, 48][ri(0, 8)]
        T = ri(max(K, 1500), 12000)
        L = max(hop, (2048 - K + 1) // hop * hop)
        nblk = -(-T // L)
        B = min(512, max(2, -(-ri(330, 700) // nblk)))
        out.append((F, K, hop, T, B, ri(0, 1) == 1, 1000 * seed + i))
    return out


def run(case, dev):
    F, K, hop, T, B, pcen, s = case
    g = torch.Generator().manual_seed(s)
    x = torch.randn(B, T, generator=g).to(dev)
    kern = torch.stack([0.1 + torch.rand(F, generator=g) * 2.9, 3.0 + torch.rand(F, generator=g) * K / 4], dim=1).to(dev)
    pw, pb = torch.full((F,), 0.4, device=dev), torch.ones(F, device=dev)
    pc = [torch.full((F,), v, device=dev) for v in (0.96, 2.0, 2.0, 0.04)]
    TP = (T - 1) // hop + 1
    go = torch.randn(B, F, TP, generator=g).to(dev)
    return _native.leaf_backward(x, kern, pw, pb, *pc, K, hop, go, pcen=pcen, need_dx=not BWD4K)


if __name__ == "__main__":
    if not os.path.exists(VARIANT):
        sys.exit(f"{VARIANT} missing: python tools/compare_builds.py --build-only cur:-DLEAF_TOOLS=1")
    _native.LIB_PATH = VARIANT
    dev = torch.device("cuda:0")
    if len(sys.argv) > 1 and sys.argv[1] == "--ref":             # child: the reference kernels, results to a file
        n, seed, path = int(sys.argv[2]), int(sys.argv[3]), sys.argv[4]
        torch.save([[t.cpu() if t is not None else None for t in run(c, dev)] for c in cases(n, seed)], path)
        sys.exit(0)
    with tempfile.TemporaryDirectory() as tmp:
        ref_path = os.path.join(tmp, "ref.pt")
        env = dict(os.environ, LEAF_WGG_BWD_DX="0", LEAF_WG_BWD_DX="0", LEAF_4K_BWD_STATIC="0")
        subprocess.run([sys.executable, os.path.abspath(__file__), "--ref", str(N), str(SEED), ref_path], check=True, env=env)
        ref = torch.load(ref_path)
    worst = 0.0
    for c, r in zip(cases(N, SEED), ref):
        a = run(c, dev)
        for _ in range(2):
            b = run(c, dev)
            for ta, tb in zip(a, b):
                if ta is not None and not torch.equal(ta, tb):
                    sys.exit(f"NOT bit-identical run to run: {c}")
        for name, ta, tr in zip(("kernel", "pool_w", "pool_b", "alpha", "delta", "root", "ema_w", "dx"), a, r):
            if ta is None:
                continue
            scale = float(tr.abs().max()) + 1e-20
            err = float((ta.cpu().double() - tr.double()).abs().max()) / scale
            worst = max(worst, err)
            # (a tensor of one or two entries is a single cancellation-prone sum: its own value is no scale for its error)
            if not err < (2e-5 if tr.numel() >= 8 else 1e-3):
                sys.exit(f"{name}: {err:.3e} of its max apart from the reference kernels: {c}")
    print(f"stress_dx{' bwd4k' if BWD4K else ''}: {N} cases (seed {SEED}), three runs each bit-identical, worst distance to the "
          f"{'static 2048-sample backward' if BWD4K else 'reference (block-per-wave / staged) kernels'} {worst:.2e} of a tensor's max")
