"""Debug (GPU box): band windows that cross Nyquist -- per-filter, per-frame error against the fp64 oracle for filters at / near the clamp mu = pi,
clip lengths with and without regular frames, several signals.  Usage: python tools/dbg_cross.py"""
import math, os, sys, torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "tests"))
import test_gpu_band as tb
from helpers import make_leaf
from oracle import leaf_oracle as lo
from leaf_pytorch_amd import _native
DEV = "cuda:0"
F = 8
mu = torch.tensor([math.pi, 3.122, 3.10, 3.05, 3.0, 2.9, 2.8, 2.0])
for sg_v in (15.3, 20.0, 30.0):
    sg = torch.full((F,), sg_v)
    geo = lo.LeafGeometry(F, 0, 401, 160, *lo.same_padding(401))
    for T in (401, 1700, 8000):
        n = torch.arange(T, dtype=torch.float64)
        gen = torch.Generator().manual_seed(3)
        sigs = {"uniform": 2 * torch.rand(1, T, generator=gen, dtype=torch.float64) - 1,
                "tone 950": torch.sin(2 * math.pi * 950 / 2048 * n).reshape(1, T),
                "tone 1000": torch.sin(2 * math.pi * 1000.4 / 2048 * n).reshape(1, T),
                "clicks": torch.zeros(1, T, dtype=torch.float64)}
        sigs["clicks"][0, torch.randint(0, T, (12,), generator=gen)] = 1.0
        for pw, bias in ((0.5, 0.1), (0.4, 1.0)):
            params = lo.default_params(geo, False, kernel=torch.stack([mu, sg], 1))
            params["_pooling.weights"] = torch.full_like(params["_pooling.weights"], pw)
            params["_pooling._bias"] = torch.full((F,), bias)
            m = make_leaf(F, 401, 160, False, params, DEV)
            cls = _native.band_classes(torch.stack([mu, sg], 1).to(DEV), torch.full((F,), pw, device=DEV), 401, 160, params["_pooling._bias"].to(DEV)).cpu().tolist()
            for name, s in sigs.items():
                x = s.float().unsqueeze(1)
                ref = lo.leaf_forward(x, params, geo, False, torch.float64)
                band, full = tb.run(m, x, tb.WG), tb.run(m, x, tb.WG | tb.FULL)
                e = ((band.double() - ref).abs() / ref.abs())[0]
                ef = ((full.double() - ref).abs() / ref.abs())[0]
                worst = e.amax(dim=1)
                if float(worst.max()) > 5e-6:
                    print(f"sigma {sg_v} T {T} pool_w {pw} bias {bias} {name:10s}: " + "  ".join(f"{cls[f]}:{float(worst[f]):.1e}@{int(e[f].argmax())}" for f in range(F)) + f"   full {float(ef.max()):.1e}  frames {e.shape[1]}")
print("done")
