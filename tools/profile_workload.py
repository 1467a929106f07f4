#!/usr/bin/env python3
"""Workload for the rocprofv3 PMC passes (profiles/): a few fused forwards at BASELINE configs[1] size plus
a staged forward whose sqmod_kernel has a KNOWN byte count in the same access pattern as the fused kernel
(coalesced 4-byte-per-lane loads/stores), used to calibrate FETCH_SIZE / WRITE_SIZE as
MI355X_MICROARCH.md section HBM prescribes."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from leaf_pytorch_amd import Leaf, _native  # noqa: E402

dev = torch.device("cuda:0")
torch.manual_seed(0)
if len(sys.argv) > 1 and sys.argv[1] == "cfg2":
    # BASELINE configs[2] per GPU: 80 filters, 32 kHz, 128 x 5 s -- the 4096-sample-plan kernel (leaf_fft_wg4k_kernel)
    m2 = Leaf(n_filters=80, sample_rate=32000).eval().to(dev)
    x2 = 2 * torch.rand(128, 1, 160000, device=dev) - 1
    with torch.no_grad():
        for _ in range(5):
            m2(x2)
    torch.cuda.synchronize()
    print("done cfg2")
    sys.exit(0)
if len(sys.argv) > 1 and sys.argv[1] in ("cfg3", "cfg4"):
    # BASELINE configs[3]: PCEN off, 512 x 1 s (two clips per workgroup); configs[4] per GPU: 256 x 10 s, bf16 I/O
    pcen, B, secs, bf16 = (False, 512, 1, False) if sys.argv[1] == "cfg3" else (True, 256, 10, True)
    mc = Leaf(pcen_compression=pcen).eval().to(dev)
    xc = 2 * torch.rand(B, 1, 16000 * secs, device=dev) - 1
    if bf16:
        xc = xc.to(torch.bfloat16)
    with torch.no_grad():
        for _ in range(5):
            mc(xc)
    torch.cuda.synchronize()
    print("done", sys.argv[1])
    sys.exit(0)
m = Leaf().eval().to(dev)
for p in m.parameters():
    p.requires_grad_(False)
x = 2 * torch.rand(256, 1, 16000, device=dev) - 1
with torch.no_grad():
    if len(sys.argv) > 1 and sys.argv[1] == "stream":              # the opt-in streaming finalize of the workgroup kernel
        m._algo = _native.ALGO_FFT_WG | _native.ALGO_STREAM_FINALIZE
        for _ in range(5):
            m(x)
        torch.cuda.synchronize()
        print("done stream")
        sys.exit(0)
    for algo in (_native.ALGO_FFT_WG, _native.ALGO_FFT, _native.ALGO_MFMA):     # the fused algorithms, 5 launches each
        m._algo = algo
        for _ in range(5):
            m(x)
    m._algo = _native.ALGO_STAGED
    m(x[:64])         # sqmod_kernel: reads 64*80*16000*4 B = 327.68 MB, writes 163.84 MB
    m(x[:64])
torch.cuda.synchronize()
print("done")
