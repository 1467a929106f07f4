import math, os, random, sys, torch
REPO=os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO,"tests"))
from oracle import leaf_oracle as lo
from leaf_pytorch_amd import _native
from helpers import grad_errors
import test_gpu_backward as tbk
base, seed = int(sys.argv[1]), int(sys.argv[2])
SB = 100000*base
DEV="cuda:0"
rng = random.Random(SB + 8800 + seed); gen = torch.Generator().manual_seed(SB + 8800 + seed)
names = ["_complex_conv._kernel", "_pooling.weights", "_pooling._bias", "_compression.alpha", "_compression.delta", "_compression.root", "_compression.ema._weights"]
c = math.sqrt(2 * math.log(2)) / math.pi
for it in range(2):
    F = rng.choice([8, 16, 24]); mu = torch.rand(F, generator=gen) * (math.pi + 0.2) - 0.1; sg = 6.0 + torch.rand(F, generator=gen) * 60.0
    if rng.random() < 0.5:
        sg[0::4] = 4 * c; sg[1::4] = 401 * c; sg[2::4] = 15.0 + torch.rand(len(sg[2::4]), generator=gen) * 2.0; sg[3::4] = 44.0 + torch.rand(len(sg[3::4]), generator=gen) * 6.0
    pcen = rng.random() < 0.7
    geo = lo.LeafGeometry(F, 0, 401, 160, *lo.same_padding(401))
    params = lo.default_params(geo, pcen, kernel=torch.stack([mu, sg], dim=1))
    params["_pooling.weights"] = (0.05 + torch.rand(F, generator=gen) * 0.5).reshape(params["_pooling.weights"].shape)
    T = rng.choice([1700, 3300, 4801, 8000]); B = -(-340 // (-(-T // 1600)))
    x = torch.randn(B, 1, T, generator=gen); grad_out = torch.randn(B, F, (T - 1) // 160 + 1, generator=gen)
    ref, _, _ = tbk.oracle_grads(x, params, geo, pcen, grad_out)
    args = [params[k].to(DEV) for k in names[:3]] + ([params[k].to(DEV) for k in names[3:]] if pcen else [None] * 4)
    res = {tag: _native.leaf_backward(x.to(DEV), *args, 401, 160, grad_out.to(DEV), pcen=pcen, **kw) for tag, kw in
           (("bias-aware", {}), ("strict", dict(strict_band_classes=True)), ("full", dict(full_transforms=True)))}
    cs = _native.band_classes(args[0], args[1], 401, 160).cpu().tolist(); cr = _native.band_classes(args[0], args[1], 401, 160, args[2]).cpu().tolist()
    print(f"case {it}: F {F} T {T} B {B} pcen {pcen}")
    r = ref["_complex_conv._kernel"].double()
    for tag, g in res.items():
        gk = g[0].cpu().double()
        rows = grad_errors("_complex_conv._kernel", gk, r)
        print(f"   {tag:10s} " + "  ".join(f"{l.split('.')[-1]} col {a:.2e} entry {b:.2f}" for l, a, b in rows))
    gk = res["bias-aware"][0].cpu().double(); gs = res["strict"][0].cpu().double(); gf = res["full"][0].cpu().double()
    top = float(r[:, 1].abs().max())
    for f in range(F):
        e = abs(float(gk[f, 1] - r[f, 1])); es = abs(float(gs[f,1]-r[f,1])); ef = abs(float(gf[f,1]-r[f,1]))
        if cs[f] != cr[f] or e > 2e-5 * top or e > 0.5*(1e-3*abs(float(r[f,1]))+1e-6*top):
            print(f"      f {f:2d} mu {float(mu[f]):.3f} sigma {float(sg[f]):.1f} pool_w {float(params['_pooling.weights'].reshape(-1)[f]):.3f} class {cs[f]}->{cr[f]}: dsigma ref {float(r[f,1]):.3e}  err bias-aware {e:.2e} strict {es:.2e} full {ef:.2e}  (col max {top:.2e})")
