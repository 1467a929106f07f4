#!/usr/bin/env python3
"""Ablation harness for the fused kernel (runs on the GPU box): builds variants of csrc/leaf_kernels.hip with
-DLEAF_ABLATE / -DLEAF_WAVES_PER_WG, and reports the HIP-event time of the fused kernel for each at BASELINE
configs[1] size.  Ablated variants produce WRONG outputs by design; this is a where-does-the-time-go tool."""
import ctypes
import os
import statistics
import subprocess
import sys

import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from leaf_pytorch_amd import _native  # noqa: E402
from leaf_pytorch_amd.initializers import GaborInit  # noqa: E402

SRC = os.path.join(REPO, "leaf_pytorch_amd", "csrc", "leaf_kernels.hip")
VARIANTS = [("baseline", []), ("no_epilogue", ["-DLEAF_ABLATE=1"]), ("no_xstage", ["-DLEAF_ABLATE=2"]),
            ("no_epi_no_x_no_store", ["-DLEAF_ABLATE=7"]), ("4waves", ["-DLEAF_WAVES_PER_WG=4"]),
            ("4waves_no_epi_x_store", ["-DLEAF_ABLATE=7", "-DLEAF_WAVES_PER_WG=4"])]
extra = sys.argv[1:]
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
libs = []
for name, flags in VARIANTS:
    so = f"/tmp/leaf_abl_{name}.so"
    subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared",
                    "-I", os.path.join(REPO, "include"), SRC, "-o", so] + flags + extra, check=True)
    lib = ctypes.CDLL(so)
    lib.leaf_workspace_bytes.restype = ctypes.c_size_t
    libs.append((name, lib))
ws = torch.empty(libs[0][1].leaf_workspace_bytes(B, T, F, K, hop, 2), dtype=torch.uint8, device=dev)
res = {n: [] for n, _ in libs}
ms = (ctypes.c_float * 3)()
for rnd in range(6):
    for name, lib in libs:
        for _ in range(3):
            rc = lib.leaf_forward_profiled_f32(P(x), B, T, P(kern), P(pw), P(pb), P(al), P(de), P(ro), P(ew), F, K, hop, 1, 2,
                                               P(out), P(ws), ctypes.c_size_t(ws.numel()), None, ms)
            assert rc == 0, (name, rc)
            if rnd:
                res[name].append(ms[1])
for name, _ in libs:
    v = res[name]
    print(f"{name:28s} fused median {statistics.median(v):.4f} ms  min {min(v):.4f}")
