"""Debug (GPU box): error of a 256-point band task whose window is centred ON Nyquist, for one full-scale tone d bins below Nyquist (its mirror
d bins above is inside the window too: a coherent pair 2 d bins apart), on the clip's first (edge) frame.  Also the two-tone case of an
ordinary window (tones 2 d apart around the centre, amplitude 0.5 each)."""
import math, os, sys, torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "tests"))
import test_gpu_band as tb
from helpers import make_leaf
from oracle import leaf_oracle as lo
from leaf_pytorch_amd import _native
DEV = "cuda:0"
F = 8
T = 1700
n = torch.arange(T, dtype=torch.float64)
for sg_v in (15.3, 18.0):
    mu = torch.tensor([math.pi, 1.8408] + [1.0] * 6)
    sg = torch.full((F,), sg_v)
    geo = lo.LeafGeometry(F, 0, 401, 160, *lo.same_padding(401))
    params = lo.default_params(geo, False, kernel=torch.stack([mu, sg], 1))
    params["_pooling.weights"] = torch.full_like(params["_pooling.weights"], 0.5)
    params["_pooling._bias"] = torch.full((F,), 0.1)
    m = make_leaf(F, 401, 160, False, params, DEV)
    print("sigma", sg_v, "classes", _native.band_classes(torch.stack([mu, sg], 1).to(DEV), torch.full((F,), 0.5, device=DEV), 401, 160, params["_pooling._bias"].to(DEV)).cpu().tolist())
    sk = 2048 / (2 * math.pi * sg_v)
    for d in range(16, 129, 8):
        x1 = torch.sin(2 * math.pi * (1024 - d + 0.3) / 2048 * n).reshape(1, 1, T).float()
        x2 = (0.5 * torch.sin(2 * math.pi * (600 - d + 0.3) / 2048 * n) + 0.5 * torch.sin(2 * math.pi * (600 + d) / 2048 * n + 1.0)).reshape(1, 1, T).float()
        out = []
        for x, f in ((x1, 0), (x2, 1)):
            ref = lo.leaf_forward(x, params, geo, False, torch.float64)
            band = tb.run(m, x, tb.WG)
            e = ((band.double() - ref).abs() / ref.abs())[0, f]
            out.append(f"{float(e[0]):.1e} / {float(e[1:-1].max()):.1e} / {float(e[-1]):.1e} (pooled {float(ref[0, f, 0] - 0.1):.1e})")
        print(f"   d {d:3d} (pair {2 * d:3d} bins apart, R^2 at d {math.exp(-(d / sk) ** 2):.1e}): window on Nyquist, one tone: first / inner / last frame {out[0]}   |   ordinary window, two tones: {out[1]}")
