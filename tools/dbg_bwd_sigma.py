import math, os, sys, torch
REPO=os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO,"tests"))
from oracle import leaf_oracle as lo
from leaf_pytorch_amd import _native
import test_gpu_backward as tbk
DEV="cuda:0"
names = ["_complex_conv._kernel", "_pooling.weights", "_pooling._bias", "_compression.alpha", "_compression.delta", "_compression.root", "_compression.ema._weights"]
for pw, T in ((0.16, 3300), (0.4, 3300), (0.16, 8000), (0.05, 3300)):
    F=24
    gen=torch.Generator().manual_seed(5)
    sg=torch.linspace(6.0, 17.5, F)
    mu=torch.full((F,), 1.5)
    geo = lo.LeafGeometry(F, 0, 401, 160, *lo.same_padding(401))
    params = lo.default_params(geo, True, kernel=torch.stack([mu, sg], dim=1))
    params["_pooling.weights"] = torch.full_like(params["_pooling.weights"], pw)
    B = -(-340 // (-(-T // 1600)))
    x = torch.randn(B, 1, T, generator=gen); go = torch.randn(B, F, (T - 1) // 160 + 1, generator=gen)
    ref, _, _ = tbk.oracle_grads(x, params, geo, True, go)
    args=[params[k].to(DEV) for k in names]
    band=_native.leaf_backward(x.to(DEV), *args, 401, 160, go.to(DEV), pcen=True, strict_band_classes=('--strict' in sys.argv))
    full=_native.leaf_backward(x.to(DEV), *args, 401, 160, go.to(DEV), pcen=True, full_transforms=True)
    cls=_native.band_classes(args[0], args[1], 401, 160).cpu().tolist()
    r=ref["_complex_conv._kernel"].double(); gb=band[0].cpu().double(); gf=full[0].cpu().double()
    print(f"pool_w {pw} T {T}: col max dmu {float(r[:,0].abs().max()):.2e} dsigma {float(r[:,1].abs().max()):.2e}")
    for f in range(F):
        print(f"   sigma {float(sg[f]):5.1f} class {cls[f]:4d}: dsigma ref {float(r[f,1]):+.3e} band err {abs(float(gb[f,1]-r[f,1])):.1e} ({abs(float(gb[f,1]-r[f,1]))/abs(float(r[f,1])):.1e} rel) full err {abs(float(gf[f,1]-r[f,1])):.1e} | dmu ref {float(r[f,0]):+.3e} band err {abs(float(gb[f,0]-r[f,0])):.1e} full {abs(float(gf[f,0]-r[f,0])):.1e}")
