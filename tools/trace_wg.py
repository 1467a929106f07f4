#!/usr/bin/env python3
"""Phase timeline of the static workgroup kernel (GPU box): -DLEAF_TRACE=1 build, s_memtime stamps of waves 0..7 of workgroup 0
over their first tasks.  Columns are ticks of s_memtime: shader-clock cycles (~0.5-0.6 ns each under this load: 10.4 k per 6 us filter task)."""
import ctypes, os, subprocess, sys
import torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from leaf_pytorch_amd.initializers import GaborInit  # noqa: E402
from leaf_pytorch_amd import _native  # noqa: E402
LEVEL = os.environ.get("LEAF_TRACE_LEVEL", "1")   # 2: task starts and finalize stamps only (the whole launch fits the buffer)
so = _native.build(variant="trace" + LEVEL, extra_flags=" ".join(["-DLEAF_TRACE=" + LEVEL, "-DLEAF_TOOLS=1"] + sys.argv[1:]))
lib = ctypes.CDLL(so); lib.leaf_workspace_bytes.restype = ctypes.c_size_t
dev = torch.device("cuda:0")
B, T, F, K, hop = int(os.environ.get("LEAF_TRACE_B", "256")), int(os.environ.get("LEAF_TRACE_T", "16000")), 40, 401, 160
ALGO = 4 | int(os.environ.get("LEAF_TRACE_ALGO_BITS", "0"), 0)      # e.g. 0x2000000 = LEAF_ALGO_STREAM_FINALIZE
torch.manual_seed(0)
x = 2 * torch.rand(B, T, device=dev) - 1
kern = GaborInit(default_window_len=K, sample_rate=16000, min_freq=60.0, max_freq=7800.0)((F, 2)).to(dev)
pw = torch.full((F,), 0.4, device=dev); pb = torch.ones(F, device=dev)
al = torch.full((F,), 0.96, device=dev); de = torch.full((F,), 2.0, device=dev)
ro = torch.full((F,), 2.0, device=dev); ew = torch.full((F,), 0.04, device=dev)
out = torch.empty(B, F, (T - 1) // hop + 1, device=dev)
P = lambda t: ctypes.c_void_p(t.data_ptr())
n = lib.leaf_workspace_bytes(B, T, F, K, hop, ALGO)
ws = torch.zeros(n, dtype=torch.uint8, device=dev)
for _ in range(20):
    assert lib.leaf_forward_f32(P(x), B, T, P(kern), P(pw), P(pb), P(al), P(de), P(ro), P(ew), F, K, hop, 1, ALGO, P(out), P(ws),
                                ctypes.c_size_t(n), None) == 0
torch.cuda.synchronize()
scales = ((B + 63) // 64) * 64 * 4                       # the per-clip scales of LEAF_FLAG_PEAKNORM sit behind the trace
tr = ws[-16 * 64 * 8 - scales: -scales].view(torch.int64).cpu().reshape(16, 64)
names = {1: "take:fwd", 2: "take:filter", 3: "spectrum-ready", 4: "multiply", 5: "transform", 6: "energies+row", 7: "pooling",
         10: "FIN-wait", 8: "FIN-start", 9: "FIN-done", 11: "band:first-transforms", 12: "band:transposed", 13: "band:pooled"}
base = min(int(v) & ((1 << 56) - 1) for v in tr[:, 0])
for w in range(16):
    row = [(int(v) >> 56, int(v) & ((1 << 56) - 1)) for v in tr[w] if int(v)]
    if not row:
        continue
    print(f"wave {w}:")
    prev = None
    line = []
    for tag, t in row[:64]:
        if tag in (1, 2):
            if line:
                print("   " + "  ".join(line))
            line = [f"@{t - base:6d} {names[tag]}"]
        else:
            line.append(f"{names.get(tag, tag)} +{t - prev}")
        prev = t
    if line:
        print("   " + "  ".join(line))
