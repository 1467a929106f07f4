#!/usr/bin/env python3
"""Phase timeline of the FFT kernel (GPU box): -DLEAF_TRACE=1 build, s_memtime stamps of waves 0 and 4 of workgroup 0."""
import ctypes, os, subprocess, sys
import torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from leaf_pytorch_amd.initializers import GaborInit  # noqa: E402
SRC = os.path.join(REPO, "leaf_pytorch_amd", "csrc", "leaf_kernels.hip")
so = "/tmp/leaf_trace_fft.so"
subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-DLEAF_TRACE=1",
                "-I", os.path.join(REPO, "include"), SRC, "-o", so] + sys.argv[1:], check=True)
lib = ctypes.CDLL(so); lib.leaf_workspace_bytes.restype = ctypes.c_size_t
dev = torch.device("cuda:0")
B, T, F, K, hop = 256, 16000, 40, 401, 160
torch.manual_seed(0)
x = 2 * torch.rand(B, T, device=dev) - 1
kern = GaborInit(default_window_len=K, sample_rate=16000, min_freq=60.0, max_freq=7800.0)((F, 2)).to(dev)
pw = torch.full((F,), 0.4, device=dev); pb = torch.ones(F, device=dev)
al = torch.full((F,), 0.96, device=dev); de = torch.full((F,), 2.0, device=dev)
ro = torch.full((F,), 2.0, device=dev); ew = torch.full((F,), 0.04, device=dev)
out = torch.empty(B, F, 100, device=dev)
P = lambda t: ctypes.c_void_p(t.data_ptr())
n = lib.leaf_workspace_bytes(B, T, F, K, hop, 3)
ws = torch.zeros(n, dtype=torch.uint8, device=dev)
for _ in range(3):
    assert lib.leaf_forward_f32(P(x), B, T, P(kern), P(pw), P(pb), P(al), P(de), P(ro), P(ew), F, K, hop, 1, 3, P(out), P(ws),
                                ctypes.c_size_t(n), None) == 0
torch.cuda.synchronize()
tr = ws[-8 * 64 * 8:].view(torch.int64).cpu().reshape(8, 64)
for w in (0, 4):
    row = [int(v) for v in tr[w]]
    t0 = row[0]
    print(f"wave {w}: task start 0, forward FFT done {row[1]-t0}")
    i = 2
    for f in range(10):
        a, b_, c = row[i], row[i + 1], row[i + 2]
        prev = row[i - 1]
        print(f"   filter {f}: Zmult {a-prev:6d}  fft {b_-a:6d}  pool+reduce+store {c-b_:6d}")
        i += 3
    print(f"   task total {row[i-1]-t0}  (next task starts at {row[i]-t0})")
