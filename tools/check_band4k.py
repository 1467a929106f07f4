#!/usr/bin/env python3
"""GPU check of the band tasks on 4096-sample blocks (32 kHz window K = 801 / hop = 320; leaf_fft_wg4k.hpp + leaf_band.hpp): the
workgroup kernel with and without them against the fp64 oracle over clip lengths that move the edge frames around, the classes
the device decides, then the timing of BASELINE configs[2] (80 filters, 128 clips of 5 s) both ways.
   usage: check_band4k.py [--quick]"""
import os
import sys
import time

import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from leaf_pytorch_amd import Leaf, _native  # noqa: E402
from oracle import leaf_oracle as lo  # noqa: E402

DEV = "cuda:0"
WG, FULL = _native.ALGO_FFT_WG, _native.ALGO_FULL_TRANSFORMS


def run(model, x, algo):
    model._algo = algo
    with torch.no_grad():
        out = model(x)
    torch.cuda.synchronize()
    return out.double().cpu()


def main():
    _native.load()
    torch.manual_seed(0)
    worst = 0.0
    for F, pcen in ((80, True), (40, True), (80, False)):
        model = Leaf(n_filters=F, sample_rate=32000, pcen_compression=pcen).eval().to(DEV)
        params = {k: v.cpu() for k, v in model.state_dict().items()}
        geo = lo.geometry(F, 32000)
        cls = _native.band_classes(model._complex_conv._kernel.detach(), model._pooling.weights.detach(), 801, 320)
        c = cls.cpu().tolist()
        print(f"F {F} pcen {pcen}: classes {({k: c.count(k) for k in sorted(set(c))})}", flush=True)
        cases = [(2, 32000, 2), (3, 32000, 0), (1, 31999, 1), (2, 32001, 2), (2, 6400, 2), (2, 3400, 0), (2, 1601, 2), (4, 801, 4),
                 (2, 160000, 2), (5, 35201, 7), (3, 3200, 3), (2, 3201, 2)]
        if not pcen:
            cases = cases[:4]
        for B, T, cus in cases:
            x = 2 * torch.rand(B, 1, T) - 1
            ref = lo.leaf_forward(x, params, geo, pcen, torch.float64)
            wg = WG | (_native.algo_reserve_cus(256 - cus) if cus else 0)
            o_band = run(model, x.to(DEV), wg)
            o_full = run(model, x.to(DEV), wg | FULL)
            eb = ((o_band - ref).abs() / ref.abs()).amax(dim=(0, 2))
            ef = ((o_full - ref).abs() / ref.abs()).amax(dim=(0, 2))
            d = ((o_band - o_full).abs() / ref.abs())
            fr = d.amax(dim=(0, 1))
            worst = max(worst, float(eb.max()))
            print(f"  B {B} T {T:6d} cus {cus}: band vs oracle {float(eb.max()):.2e} (filter {int(eb.argmax())})  full vs oracle {float(ef.max()):.2e}  "
                  f"band vs full {float(d.max()):.2e} at frame {int(fr.argmax())} of {ref.shape[-1]}; finite {bool(torch.isfinite(o_band).all())} "
                  f"differ {not torch.equal(o_band, o_full)}", flush=True)
    print(f"worst band vs oracle {worst:.2e}")
    if "--quick" in sys.argv:
        return
    model = Leaf(n_filters=80, sample_rate=32000).eval().to(DEV)
    x = (2 * torch.rand(128, 1, 160000) - 1).to(DEV)
    for name, algo in (("band", WG), ("full", WG | FULL), ("band", WG), ("full", WG | FULL)):
        model._algo = algo
        with torch.no_grad():
            for _ in range(20):
                model(x)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(50):
                model(x)
            torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / 50 * 1e3
        print(f"cfg2 {name}: {ms:.4f} ms per call, {128 * 500 / ms / 1e3:.1f} M frames/s", flush=True)
    params = {k: v.cpu() for k, v in model.state_dict().items()}
    ref = lo.leaf_forward(x[:2].cpu(), params, lo.geometry(80, 32000), True, torch.float64)
    o = run(model, x, WG)
    print(f"cfg2 band, first 2 clips vs oracle: {float(((o[:2] - ref).abs() / ref.abs()).max()):.2e}")


if __name__ == "__main__":
    main()
