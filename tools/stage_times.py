import sys, torch
sys.path.insert(0, '/root/repo')
from leaf_pytorch_amd import Leaf, _native
dev = torch.device('cuda:0')
m = Leaf().eval().to(dev)
sd = m.state_dict()
prm = (sd["_complex_conv._kernel"], sd["_pooling.weights"], sd["_pooling._bias"], sd["_compression.alpha"],
       sd["_compression.delta"], sd["_compression.root"], sd["_compression.ema._weights"])
for B, T in ((4, 16000), (1, 16000), (16, 16000), (24, 16000), (300, 16000), (256, 16000), (8, 160000)):
    x = torch.randn(B, 1, T, device=dev)
    for _ in range(200): _native.leaf_forward_profiled(x, *prm, 401, 160)
    acc = [0, 0, 0]
    for _ in range(50):
        _, ms = _native.leaf_forward_profiled(x, *prm, 401, 160)
        acc = [a + b for a, b in zip(acc, ms)]
    print(B, T, [round(a / 50 * 1e3, 1) for a in acc], "us (prep, main, finalize)")
