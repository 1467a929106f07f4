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
    SF = _native.ALGO_STREAM_FINALIZE
    # (B, T, CUs the call may use (0: all), extra algo bits, what it exercises)
    cases = [(2, 16000, 2, 0, "one clip per workgroup: sums in LDS"), (3, 16000, 3, 0, "sums in LDS"), (1, 15999, 1, 0, "sums in LDS"),
             (2, 16001, 2, 0, "sums in LDS, 11th block of one sample"), (2, 8000, 2, 0, "sums in LDS"), (2, 3200, 2, 0, "sums in LDS"),
             (2, 1700, 2, 0, "sums in LDS"), (2, 16160, 2, 0, "sums in LDS"), (2, 801, 2, 0, "every frame an edge frame"),
             (4, 401, 4, 0, "one block, three frames"),
             (3, 16000, 0, 0, "one block per workgroup: partial sums in HBM + row kernel"), (5, 16001, 7, 0, "clips straddle workgroups"),
             (2, 16000, 2, SF, "streaming finalize"), (4, 16000, 2, SF, "streaming finalize, two clips per workgroup"),
             (2, 160000, 2, 0, "10 s clips: streaming finalize (auto)"), (3, 47999, 3, SF, "streaming finalize, 3 s"),
             (2, 16160, 1, SF, "streaming finalize, two clips on one workgroup")]
    for B, T, cus, bits, what in cases:
        x = 2 * torch.rand(B, 1, T) - 1
        ref = lo.leaf_forward(x, params, geo, True, torch.float64)
        wg = _native.ALGO_FFT_WG | bits | (_native.algo_reserve_cus(256 - cus) if cus else 0)
        o_band = run(model, x.to(DEV), wg)
        o_full = run(model, x.to(DEV), wg | _native.ALGO_FULL_TRANSFORMS)
        eb = ((o_band - ref).abs() / ref.abs()).amax(dim=(0, 2))
        ef = ((o_full - ref).abs() / ref.abs()).amax(dim=(0, 2))
        d = ((o_band - o_full).abs() / ref.abs())
        fr = d.amax(dim=(0, 1))
        worst = max(worst, float(eb.max()))
        print(f"B {B} T {T:6d} [{what}]: band vs oracle {float(eb.max()):.2e} (filter {int(eb.argmax())})  full vs oracle {float(ef.max()):.2e}  "
              f"band vs full {float(d.max()):.2e} at frame {int(fr.argmax())} of {ref.shape[-1]}; finite {bool(torch.isfinite(o_band).all())}", flush=True)
    # PCEN off (BASELINE configs[3])
    m3 = Leaf(pcen_compression=False).eval().to(DEV)
    p3 = {k: v.cpu() for k, v in m3.state_dict().items()}
    for B, T, cus, bits in ((2, 16000, 2, 0), (4, 16000, 2, 0), (3, 16000, 0, 0)):
        x = 2 * torch.rand(B, 1, T) - 1
        ref = lo.leaf_forward(x, p3, geo, False, torch.float64)
        wg = _native.ALGO_FFT_WG | bits | (_native.algo_reserve_cus(256 - cus) if cus else 0)
        o_band = run(m3, x.to(DEV), wg)
        o_full = run(m3, x.to(DEV), wg | _native.ALGO_FULL_TRANSFORMS)
        print(f"PCEN off B {B} T {T}: band vs oracle {float(((o_band - ref).abs() / ref.abs()).max()):.2e}  full vs oracle "
              f"{float(((o_full - ref).abs() / ref.abs()).max()):.2e}  band vs full {float(((o_band - o_full).abs() / ref.abs()).max()):.2e}", flush=True)
        worst = max(worst, float(((o_band - ref).abs() / ref.abs()).max()))
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
    for name, mdl, Bc, Tc in (("cfg3", m3, 512, 16000), ("cfg4", model, 256, 160000)):
        xc = (2 * torch.rand(Bc, 1, Tc) - 1).to(DEV)
        for tag, algo in (("band", _native.ALGO_FFT_WG), ("full", _native.ALGO_FFT_WG | _native.ALGO_FULL_TRANSFORMS)):
            mdl._algo = algo
            with torch.no_grad():
                for _ in range(30):
                    mdl(xc)
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(60):
                    mdl(xc)
                torch.cuda.synchronize()
            ms = (time.perf_counter() - t0) / 60 * 1e3
            print(f"{name} {tag}: {ms:.4f} ms per call, {Bc * ((Tc - 1) // 160 + 1) / ms / 1e3:.1f} M frames/s", flush=True)
        del xc
    ref = lo.leaf_forward(x[:4].cpu(), params, geo, True, torch.float64)
    o = run(model, x, _native.ALGO_FFT_WG)
    print(f"cfg1 band, first 4 clips vs oracle: {float(((o[:4] - ref).abs() / ref.abs()).max()):.2e}")


if __name__ == "__main__":
    main()
