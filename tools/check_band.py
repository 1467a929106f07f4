#!/usr/bin/env python3
"""GPU check of the band-limited filter tasks (leaf_band.hpp): the workgroup kernel with and without them against the fp64
oracle, per filter, over clip lengths that move the edge frames around, then the timing of BASELINE configs[1] both ways.
   usage: check_band.py [--quick]"""
import os
import sys
import time

import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from leaf_pytorch_amd import Leaf, _native  # noqa: E402
from oracle import leaf_oracle as lo  # noqa: E402

DEV = "cuda:0"


def run(model, x, algo):
    model._algo = algo
    with torch.no_grad():
        out = model(x)
    torch.cuda.synchronize()
    return out.double().cpu()


def main():
    _native.load()
    torch.manual_seed(0)
    model = Leaf().eval().to(DEV)
    params = {k: v.cpu() for k, v in model.state_dict().items()}
    geo = lo.geometry()
    worst = 0.0
    for B, T in ((2, 16000), (3, 16000), (1, 15999), (2, 16001), (2, 8000), (2, 3200), (2, 1700), (1, 48000), (2, 16160), (2, 801), (4, 401)):
        x = 2 * torch.rand(B, 1, T) - 1
        ref = lo.leaf_forward(x, params, geo, True, torch.float64)
        wg = _native.ALGO_FFT_WG | _native.algo_reserve_cus(256 - B)           # one clip per workgroup: frame sums in LDS
        o_band = run(model, x.to(DEV), wg)
        o_full = run(model, x.to(DEV), wg | _native.ALGO_FULL_TRANSFORMS)
        eb = ((o_band - ref).abs() / ref.abs()).amax(dim=(0, 2))
        ef = ((o_full - ref).abs() / ref.abs()).amax(dim=(0, 2))
        d = ((o_band - o_full).abs() / ref.abs())
        fr = d.amax(dim=(0, 1))
        worst = max(worst, float(eb.max()))
        print(f"B {B} T {T:6d}: band vs oracle {float(eb.max()):.2e} (filter {int(eb.argmax())})  full vs oracle {float(ef.max()):.2e}  "
              f"band vs full {float(d.max()):.2e} at frame {int(fr.argmax())} of {ref.shape[-1]}; finite {bool(torch.isfinite(o_band).all())}")
        if T == 16000 and B == 2:
            print("   per filter band vs oracle:", " ".join(f"{float(v):.0e}" for v in eb))
    print(f"worst band vs oracle {worst:.2e}")
    if "--quick" in sys.argv:
        return
    # timing at BASELINE configs[1]
    x = (2 * torch.rand(256, 1, 16000) - 1).to(DEV)
    for name, algo in (("band", _native.ALGO_FFT_WG), ("full", _native.ALGO_FFT_WG | _native.ALGO_FULL_TRANSFORMS), ("band", _native.ALGO_FFT_WG),
                       ("full", _native.ALGO_FFT_WG | _native.ALGO_FULL_TRANSFORMS)):
        model._algo = algo
        with torch.no_grad():
            for _ in range(300):
                model(x)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(500):
                model(x)
            torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / 500 * 1e3
        print(f"cfg1 {name}: {ms:.4f} ms per call, {256 * 100 / ms / 1e3:.1f} M frames/s")
    ref = lo.leaf_forward(x[:4].cpu(), params, geo, True, torch.float64)
    o = run(model, x, _native.ALGO_FFT_WG)
    print(f"cfg1 band, first 4 clips vs oracle: {float(((o[:4] - ref).abs() / ref.abs()).max()):.2e}")


if __name__ == "__main__":
    main()
