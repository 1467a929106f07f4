import math, sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from leaf_pytorch_amd import Leaf, _native
from oracle import leaf_oracle as lo
DEV="cuda:0"
WG, FULL, STRICT = _native.ALGO_FFT_WG, _native.ALGO_FULL_TRANSFORMS, _native.ALGO_STRICT_BAND_CLASSES
def run(m,x,a):
    m._algo=a
    with torch.no_grad(): o=m(x.to(DEV))
    return o.double().cpu()
model=Leaf().eval().to(DEV)
N,K=2048,401
T=3*(N-K+1)-7
n=torch.arange(T,dtype=torch.float64)
a=WG|_native.algo_reserve_cus(254)
for bias in (1.0, 1e-3):
    with torch.no_grad(): model._pooling._bias.fill_(bias)
    params={k:v.cpu() for k,v in model.state_dict().items()}
    for kt in (202.0, 60.0, 330.5):
        x=torch.sin(2*math.pi*kt/N*n).float().reshape(1,1,T).repeat(2,1,1)
        ref=lo.leaf_forward(x,params,lo.geometry(),True,torch.float64)
        # pooled energies per filter (oracle, fp64)
        p64={k:v.double() for k,v in params.items()}
        outs={nm:run(model,x,a|bits) for nm,bits in (("relaxed",0),("strict",STRICT),("full",FULL))}
        print(f"bias {bias} tone bin {kt}:")
        for f in (5,6,7,9,10,11,20,38):
            r=ref[:,f]
            print(f"   filter {f:2d}: " + "  ".join(f"{nm} {float(((o[:,f]-r).abs()/r.abs()).max()):.2e}" for nm,o in outs.items()) + f"   out range {float(r.min()):.3e}..{float(r.max()):.3e}")
