#!/usr/bin/env python3
"""Phase timeline of the fused kernel (GPU box): builds a -DLEAF_TRACE=1 variant, runs BASELINE configs[1] and
prints, for waves 0 and 4 of workgroup 0 (SIMD partners), the s_memtime stamps of the first tasks."""
import ctypes
import os
import subprocess
import sys

import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from leaf_pytorch_amd.initializers import GaborInit  # noqa: E402

SRC = os.path.join(REPO, "leaf_pytorch_amd", "csrc", "leaf_kernels.hip")
so = "/tmp/leaf_trace.so"
subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-DLEAF_TRACE=1",
                "-I", os.path.join(REPO, "include"), SRC, "-o", so] + sys.argv[1:], check=True)
lib = ctypes.CDLL(so)
lib.leaf_workspace_bytes.restype = ctypes.c_size_t
dev = torch.device("cuda:0")
B, T, F, K, hop = 256, 16000, 40, 401, 160
torch.manual_seed(0)
x = (2 * torch.rand(B, T, device=dev) - 1)
kern = GaborInit(default_window_len=K, sample_rate=16000, min_freq=60.0, max_freq=7800.0)((F, 2)).to(dev)
pw = torch.full((F,), 0.4, device=dev); pb = torch.ones(F, device=dev)
al = torch.full((F,), 0.96, device=dev); de = torch.full((F,), 2.0, device=dev)
ro = torch.full((F,), 2.0, device=dev); ew = torch.full((F,), 0.04, device=dev)
out = torch.empty(B, F, 100, device=dev)
P = lambda t: ctypes.c_void_p(t.data_ptr())
nbytes = lib.leaf_workspace_bytes(B, T, F, K, hop, 2)
ws = torch.zeros(nbytes, dtype=torch.uint8, device=dev)
for _ in range(3):
    rc = lib.leaf_forward_f32(P(x), B, T, P(kern), P(pw), P(pb), P(al), P(de), P(ro), P(ew), F, K, hop, 1, 2,
                              P(out), P(ws), ctypes.c_size_t(nbytes), None)
    assert rc == 0
torch.cuda.synchronize()
tr = ws[-8 * 64 * 8:].view(torch.int64).cpu().reshape(8, 64)
t0 = int(tr[:, 0].min())
names = ["task", "staged", "k0_start", "k0_end", "epi0_end", "k1_start", "k1_end", "epi1_end"]
for w in (0, 4, 1, 5):
    row = [int(v) - t0 for v in tr[w]]
    print(f"wave {w}:")
    for i in range(0, 32, 8):
        seg = row[i:i + 8]
        print("   " + "  ".join(f"{n}={v}" for n, v in zip(names, seg)))
        print("      durations: stage %d  k0 %d  epi0 %d  k1 %d  epi1 %d  | task total %d" % (
            seg[1] - seg[0], seg[3] - seg[2], seg[4] - seg[3], seg[6] - seg[5], seg[7] - seg[6],
            (row[i + 8] - seg[0]) if i + 8 < 64 else -1))
