import math, os, random, sys, torch
REPO=os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO,"tests"))
import test_gpu_band as tb
from helpers import make_leaf
from oracle import leaf_oracle as lo
from leaf_pytorch_amd import _native
seed=int(sys.argv[1]); K=int(sys.argv[2]) if len(sys.argv)>2 else 401
DEV="cuda:0"
if K==401:
    rng = random.Random(tb.SEED_BASE + 5000 + seed); gen = torch.Generator().manual_seed(tb.SEED_BASE + 77 + seed)
    for it in range(4):
        F = rng.choice([8, 16, 40]); kernel, pool_w = tb._fuzz_params(rng, gen, F); pcen = rng.random() < 0.75
        geo = lo.LeafGeometry(F, 0, 401, 160, *lo.same_padding(401)); params = lo.default_params(geo, pcen, kernel=kernel)
        params["_pooling.weights"] = pool_w.reshape(params["_pooling.weights"].shape); params["_pooling._bias"] = tb._fuzz_bias(rng, gen, F)
        B = rng.choice([1, 2, 3]); T = rng.choice([401, 1700, 3300, 8000, 15999, 16000, 16001, 16160, 20000]); site = rng.randrange(3)
        algo = tb.WG | (tb.cus(B) if site == 0 else 0 if site == 1 else (tb.SF | tb.cus(1)))
        st=rng.getstate(); x = tb._fuzz_signal(rng, gen, B, T)
        m = make_leaf(F, 401, 160, pcen, params, DEV); ref = lo.leaf_forward(x, params, geo, pcen, torch.float64)
        band, full, strict = tb.run(m, x, algo), tb.run(m, x, algo | tb.FULL), tb.run(m, x, algo | tb.STRICT)
        e=((band.double()-ref).abs()/ref.abs()).amax(dim=(0,2)); es=((strict.double()-ref).abs()/ref.abs()).amax(dim=(0,2)); ef=((full.double()-ref).abs()/ref.abs()).amax(dim=(0,2))
        cs=_native.band_classes(kernel.to(DEV),pool_w.to(DEV),401,160).cpu(); cr=_native.band_classes(kernel.to(DEV),pool_w.to(DEV),401,160,params["_pooling._bias"].to(DEV)).cpu()
        print(f"case {it}: F {F} B {B} T {T} site {site} pcen {pcen}: worst relaxed {float(e.max()):.2e} strict {float(es.max()):.2e} full {float(ef.max()):.2e}; x absmax {float(x.abs().max()):.2f}")
        for f in range(F):
            if int(cs[f])!=int(cr[f]) or e[f]>1e-5:
                c=math.sqrt(2*math.log(2))/math.pi
                print(f"    f {f:2d} mu {float(kernel[f,0]):.3f} (bin {float(kernel[f,0].clamp(0,math.pi))*2048/6.2832:.0f}) sigma {float(kernel[f,1].clamp(4*c,401*c)):.1f} pool_w {float(pool_w[f]):.3f} bias {float(params['_pooling._bias'][f]):.3g}: class {int(cs[f])}->{int(cr[f])}  err relaxed {float(e[f]):.2e} strict {float(es[f]):.2e} full {float(ef[f]):.2e}  out range {float(ref[:,f].min()):.2e}..{float(ref[:,f].max()):.2e}")

else:
    rng = random.Random(tb.SEED_BASE + 9000 + seed); gen = torch.Generator().manual_seed(tb.SEED_BASE + 177 + seed)
    c = math.sqrt(2 * math.log(2)) / math.pi
    for it in range(3):
        F = rng.choice([8, 16, 40]); kernel, pool_w = tb._fuzz_params(rng, gen, F); kernel[:, 1] = kernel[:, 1] * 2.0
        if rng.random() < 0.5:
            kernel[0::4, 1] = 4 * c; kernel[1::4, 1] = 801 * c; kernel[2::4, 1] = 60.0 + torch.rand(len(kernel[2::4, 1]), generator=gen) * 40.0
        pcen = rng.random() < 0.75
        geo = lo.LeafGeometry(F, 0, 801, 320, *lo.same_padding(801)); params = lo.default_params(geo, pcen, kernel=kernel)
        params["_pooling.weights"] = pool_w.reshape(params["_pooling.weights"].shape); params["_pooling._bias"] = tb._fuzz_bias(rng, gen, F)
        B = rng.choice([1, 2, 3]); T = rng.choice([801, 3400, 6600, 16000, 31999, 32000, 32001, 35520])
        algo = tb.WG | (tb.cus(B) if rng.random() < 0.5 else 0)
        x = tb._fuzz_signal(rng, gen, B, T)
        m = make_leaf(F, 801, 320, pcen, params, DEV); ref = lo.leaf_forward(x, params, geo, pcen, torch.float64)
        band, full, strict = tb.run(m, x, algo), tb.run(m, x, algo | tb.FULL), tb.run(m, x, algo | tb.STRICT)
        e=((band.double()-ref).abs()/ref.abs()).amax(dim=(0,2)); es=((strict.double()-ref).abs()/ref.abs()).amax(dim=(0,2)); ef=((full.double()-ref).abs()/ref.abs()).amax(dim=(0,2))
        cs=_native.band_classes(kernel.to(DEV),pool_w.to(DEV),801,320).cpu(); cr=_native.band_classes(kernel.to(DEV),pool_w.to(DEV),801,320,params["_pooling._bias"].to(DEV)).cpu()
        print(f"case {it}: F {F} B {B} T {T} pcen {pcen}: worst relaxed {float(e.max()):.2e} strict {float(es.max()):.2e} full {float(ef.max()):.2e}; x absmax {float(x.abs().max()):.2f}")
        for f in range(F):
            if int(cs[f])!=int(cr[f]) or e[f]>1e-5:
                print(f"    f {f:2d} mu {float(kernel[f,0]):.3f} (bin {float(kernel[f,0].clamp(0,math.pi))*4096/6.2832:.0f}) sigma {float(kernel[f,1].clamp(4*c,801*c)):.1f} pool_w {float(pool_w[f]):.3f} bias {float(params['_pooling._bias'][f]):.3g}: class {int(cs[f])}->{int(cr[f])}  err relaxed {float(e[f]):.2e} strict {float(es[f]):.2e} full {float(ef[f]):.2e}  out range {float(ref[:,f].min()):.2e}..{float(ref[:,f].max()):.2e}")
