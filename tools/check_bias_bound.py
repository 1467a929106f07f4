#!/usr/bin/env python3
"""GPU check of round 6's bias-aware energy bound of the band classes (leaf_band.hpp: kBandBiasScaleMax): which filters it admits
at the default initialisations, what the newly admitted filters' outputs do against the fp64 oracle, the full-transform path and
round 5's strict decision (LEAF_ALGO_STRICT_BAND_CLASSES) over six kinds of signal -- incl. full-scale tones placed in the first
dropped side lobe of every newly admitted filter, the case the bound is about -- and the timing of the four BASELINE configs
both ways.
   usage: check_bias_bound.py [--quick]"""
import math
import os
import sys
import time

import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from leaf_pytorch_amd import Leaf, _native  # noqa: E402
from oracle import leaf_oracle as lo  # noqa: E402

DEV = "cuda:0"
WG, FULL, STRICT = _native.ALGO_FFT_WG, _native.ALGO_FULL_TRANSFORMS, _native.ALGO_STRICT_BAND_CLASSES


def run(model, x, algo):
    model._algo = algo
    with torch.no_grad():
        out = model(x)
    torch.cuda.synchronize()
    return out.double().cpu()


def signals(T, gen, tone_bins, N):
    n = torch.arange(T, dtype=torch.float64)
    out = {"uniform": 2 * torch.rand(T, generator=gen, dtype=torch.float64) - 1,
           "normal": torch.randn(T, generator=gen, dtype=torch.float64),
           "chirp": torch.sin(math.pi * n * n / (2 * T)),
           "six tones": sum(torch.sin((0.1 + 0.5 * i) * n + i) for i in range(6)) / 6}
    clicks = torch.zeros(T, dtype=torch.float64)
    clicks[torch.randint(0, T, (12,), generator=gen)] = 1.0
    out["clicks"] = clicks
    for name, k in tone_bins.items():                      # a full-scale tone in the first side lobe the window drops
        out[name] = torch.sin(2 * math.pi * k / N * n)
    return out


def main():
    _native.load()
    gen = torch.Generator().manual_seed(0)
    for sr, F, N in ((16000, 40, 2048), (32000, 80, 4096)):
        model = Leaf(n_filters=F, sample_rate=sr).eval().to(DEV)
        K, hop = model._complex_conv._kernel_size, model._pooling.strides
        sd = model.state_dict()
        strict = _native.band_classes(sd["_complex_conv._kernel"], sd["_pooling.weights"], K, hop).cpu()
        relaxed = _native.band_classes(sd["_complex_conv._kernel"], sd["_pooling.weights"], K, hop, sd["_pooling._bias"]).cpu()
        new = [f for f in range(F) if int(strict[f]) != int(relaxed[f])]
        count = lambda c: {int(v): int((c == v).sum()) for v in sorted(set(c.tolist()))}
        print(f"{sr} Hz, {F} filters: strict classes {count(strict)}  with bias 1.0 {count(relaxed)}  newly admitted: {new}")
        params = {k: v.cpu() for k, v in sd.items()}
        geo = lo.geometry(F, sr)
        mu = sd["_complex_conv._kernel"][:, 0].cpu()
        T = 5 * (N - K + 1) // 2
        # tones half a window (+ 8 bins) above and below the centre bin of each newly admitted filter: just outside its window
        M = 256 if sr == 16000 else 512
        tone_bins = {}
        for f in new[:3] + new[-2:]:
            k0 = float(mu[f]) * N / (2 * math.pi)
            Mf = int(relaxed[f])
            tone_bins[f"tone above the window of filter {f}"] = k0 + Mf / 2 + 8
            if k0 - Mf / 2 - 8 > 1:
                tone_bins[f"tone below the window of filter {f}"] = k0 - Mf / 2 - 8
        worst = 0.0
        for name, sig in signals(T, gen, tone_bins, N).items():
            x = sig.float().reshape(1, 1, T).repeat(2, 1, 1)
            ref = lo.leaf_forward(x, params, geo, True, torch.float64)
            xd = x.to(DEV)
            a = WG | _native.algo_reserve_cus(254)
            o_new, o_old, o_full = run(model, xd, a), run(model, xd, a | STRICT), run(model, xd, a | FULL)
            rel = lambda o: ((o - ref).abs() / ref.abs())
            en = rel(o_new)
            e_new_f = float(en[:, new].max()) if new else 0.0
            worst = max(worst, float(en.max()))
            print(f"   {name:42s} vs oracle: bias-aware {float(en.max()):.2e} (newly admitted filters {e_new_f:.2e})  strict {float(rel(o_old).max()):.2e}  "
                  f"full {float(rel(o_full).max()):.2e};  bias-aware vs full {float(((o_new - o_full).abs() / ref.abs()).max()):.2e}", flush=True)
        print(f"   worst bias-aware vs oracle at {sr} Hz: {worst:.2e}")
    if "--quick" in sys.argv:
        return
    cfgs = (("cfg1", dict(), 256, 16000, torch.float32), ("cfg2", dict(n_filters=80, sample_rate=32000), 128, 160000, torch.float32),
            ("cfg3", dict(pcen_compression=False), 512, 16000, torch.float32), ("cfg4", dict(), 256, 160000, torch.bfloat16))
    for name, kw, B, T, dt in cfgs:
        m = Leaf(**kw).eval().to(DEV)
        x = (2 * torch.rand(B, 1, T, device=DEV) - 1).to(dt)
        hopc = m._pooling.strides
        res = {}
        for rnd in range(3):
            for tag, algo in (("bias-aware", _native.ALGO_AUTO), ("strict", _native.ALGO_AUTO | STRICT)):
                m._algo = algo
                with torch.no_grad():
                    for _ in range(30):
                        m(x)
                    torch.cuda.synchronize()
                    n = 300 if T <= 16000 else 40
                    t0 = time.perf_counter()
                    for _ in range(n):
                        m(x)
                    torch.cuda.synchronize()
                res.setdefault(tag, []).append((time.perf_counter() - t0) / n * 1e3)
        frames = B * ((T - 1) // hopc + 1)
        print(f"{name}: bias-aware {min(res['bias-aware']):.4f} ms ({frames / min(res['bias-aware']) / 1e3:.1f} M frames/s)   strict (round 5) "
              f"{min(res['strict']):.4f} ms ({frames / min(res['strict']) / 1e3:.1f} M frames/s)", flush=True)


if __name__ == "__main__":
    main()
