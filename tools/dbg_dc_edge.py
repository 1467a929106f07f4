"""Debug (GPU box): a filter whose window starts at bin 1 (its lower tail reaches DC), weak tone in the core + strong component at / next to DC --
case 117 of tools/band_proto.py --bias-fuzz 180 --eta 2e-6 (fp64 model: 3.9e-5 of bias + pooled energy)."""
import math, os, sys, torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "tests"))
import test_gpu_band as tb
from helpers import make_leaf
from oracle import leaf_oracle as lo
from leaf_pytorch_amd import _native
DEV = "cuda:0"
F = 8
for k0, sg_v, pw, bias in ((102, 14.3, 0.4, 0.3), (102, 14.3, 0.4, 1.0), (90, 14.3, 0.4, 0.3), (120, 12.0, 0.4, 0.3), (60, 30.0, 0.4, 0.3)):
    mu = torch.full((F,), 2 * math.pi * k0 / 2048)
    sg = torch.full((F,), sg_v)
    geo = lo.LeafGeometry(F, 0, 401, 160, *lo.same_padding(401))
    params = lo.default_params(geo, False, kernel=torch.stack([mu, sg], 1))
    params["_pooling.weights"] = torch.full_like(params["_pooling.weights"], pw)
    params["_pooling._bias"] = torch.full((F,), bias)
    m = make_leaf(F, 401, 160, False, params, DEV)
    cls = _native.band_classes(torch.stack([mu, sg], 1).to(DEV), torch.full((F,), pw, device=DEV), 401, 160, params["_pooling._bias"].to(DEV)).cpu().tolist()
    for T in (1700, 5000):
        n = torch.arange(T, dtype=torch.float64)
        out = []
        for kd in (0.0, 0.5, 1.0, 2.0, 4.0):
            for a_core in (0.02, 0.0):
                x = (a_core * torch.sin(2 * math.pi * k0 / 2048 * n) + 0.98 * torch.sin(2 * math.pi * kd / 2048 * n + 1.0)).reshape(1, 1, T).float()
                ref = lo.leaf_forward(x, params, geo, False, torch.float64)
                band, full = tb.run(m, x, tb.WG), tb.run(m, x, tb.WG | tb.FULL)
                out.append(f"{kd}/{a_core}: {tb.rel_err(band[:, 0], ref[:, 0]):.1e} ({tb.rel_err(full[:, 0], ref[:, 0]):.0e})")
        print(f"bin {k0} sigma {sg_v} bias {bias} class {cls[0]} T {T}: " + "  ".join(out))
