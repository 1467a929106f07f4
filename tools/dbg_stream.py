import sys, torch
sys.path.insert(0, '/root/repo')
from leaf_pytorch_amd import Leaf, _native
dev = 'cuda:0'
torch.manual_seed(0)
x = 2 * torch.rand(256, 1, 16000, device=dev) - 1
def run(tag, **kw):
    m = Leaf().eval().to(dev)
    with torch.no_grad():
        for k, v in kw.items():
            getattr_path = {"alpha": m._compression.alpha, "delta": m._compression.delta, "root": m._compression.root,
                            "ema": m._compression.ema._weights, "bias": m._pooling._bias}[k]
            getattr_path.fill_(v)
    m._algo = _native.ALGO_FFT_WG
    with torch.no_grad():
        big = m(x)
        small = m(x[40:43])
    d = (big[40:43] - small).abs()
    print(f"{tag:30s} mismatching {int((d > 0).sum()):6d} of {d.numel()}  max rel {float((d / small.abs().clamp_min(1e-30)).max()):.3e}", flush=True)
run("default")
run("alpha=0 (q = p)", alpha=0.0)
run("ema_w=1 (M = p)", ema=1.0)
run("ema_w=0 (M = p0)", ema=0.0)
run("root=1", root=1.0)
run("alpha=0, root=1", alpha=0.0, root=1.0)
run("alpha=0, root=1, delta=1", alpha=0.0, root=1.0, delta=1.0)
run("alpha=1", alpha=1.0)
