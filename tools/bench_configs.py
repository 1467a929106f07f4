#!/usr/bin/env python3
"""Per-GPU timing of every BASELINE.json config the way SURVEY.md §8(d) specifies it (they are parity cases, not
bench.py lines): one GPU, the per-GPU batch of each config, >=10 warm-up calls (repeated for 0.2 s so that the clocks
have settled) + >=50 timed calls of the whole `Leaf.forward`, one HIP-event pair per call, median + p10/p90.  Inputs: U(-1,1) (primary) and N(0,1) (secondary),
seed 0; parameters: the constructor defaults and a seeded +-10 % perturbation (so no clamp/pow sits at its init value).
Also prints the measured device stream-copy rate (the practical HBM roof next to the nominal 8 TB/s).
One JSON line per measurement."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from leaf_pytorch_amd import Leaf, _native  # noqa: E402
import bench  # noqa: E402  (executed-flop model of the kernels and the fp32 VALU peak: the same accounting as the bench line)

dev = torch.device("cuda:0")
WARMUP, TIMED = 10, 50
ALGO_NAMES = {_native.ALGO_STAGED: "staged", _native.ALGO_MFMA: "mfma", _native.ALGO_FFT: "fft", _native.ALGO_FFT_WG: "fft_wg",
              _native.ALGO_FFT_SMALL: "fft_small"}


def timed_calls(fn):
    import time
    t0 = time.perf_counter()                              # spin-up: a fresh / idle GPU needs ~0.1 s of work to settle its clocks
    while time.perf_counter() - t0 < 0.2:
        for _ in range(WARMUP):
            fn()
        torch.cuda.synchronize()
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(TIMED)]
    for s, e in evs:
        s.record()
        fn()
        e.record()
    torch.cuda.synchronize()
    ms = sorted(s.elapsed_time(e) for s, e in evs)
    return ms[len(ms) // 2], ms[len(ms) // 10], ms[(9 * len(ms)) // 10]


def stream_copy_peak():
    n = 1 << 28                                         # 1 GiB read + 1 GiB write per copy
    a = torch.empty(n, dtype=torch.float32, device=dev).normal_()
    b = torch.empty_like(a)
    med, p10, _ = timed_calls(lambda: b.copy_(a))
    return 2 * 4 * n / med / 1e6, 2 * 4 * n / p10 / 1e6


CONFIGS = [
    # name, ctor kwargs, per-GPU batch, seconds, pcen, bf16 I/O, input distribution, perturbed parameters
    ("cfg0 default B=4x1s", dict(n_filters=40, sample_rate=16000), 4, 1.0, True, False, "uniform", False),
    ("cfg1 default B=256x1s", dict(n_filters=40, sample_rate=16000), 256, 1.0, True, False, "uniform", False),
    ("cfg1 N(0,1) input", dict(n_filters=40, sample_rate=16000), 256, 1.0, True, False, "normal", False),
    ("cfg1 perturbed parameters", dict(n_filters=40, sample_rate=16000), 256, 1.0, True, False, "uniform", True),
    ("cfg2 80f/32k/5s per-GPU B=128", dict(n_filters=80, sample_rate=32000), 128, 5.0, True, False, "uniform", False),
    ("cfg3 PCEN off B=512x1s", dict(n_filters=40, sample_rate=16000), 512, 1.0, False, False, "uniform", False),
    ("cfg4 40f/16k/10s per-GPU B=256 fp32 I/O", dict(n_filters=40, sample_rate=16000), 256, 10.0, True, False, "uniform", False),
    ("cfg4 40f/16k/10s per-GPU B=256 bf16 I/O", dict(n_filters=40, sample_rate=16000), 256, 10.0, True, True, "uniform", False),
    ("audioset-cfg 64f/16k/1s B=256", dict(n_filters=64, sample_rate=16000), 256, 1.0, True, False, "uniform", False),
]

gbps_med, gbps_best = stream_copy_peak()
print(json.dumps({"stream_copy_GBps_median": round(gbps_med), "stream_copy_GBps_p10": round(gbps_best),
                  "nominal_hbm_GBps": 8000}), flush=True)

for name, kw, B, secs, pcen, bf16, dist, perturbed in CONFIGS:
    torch.manual_seed(0)
    m = Leaf(pcen_compression=pcen, **kw).eval().to(dev)
    if perturbed:
        g = torch.Generator(device="cpu").manual_seed(1)
        with torch.no_grad():
            for p in m.parameters():
                p.mul_((1.0 + 0.1 * (2 * torch.rand(p.shape, generator=g) - 1)).to(dev))
    for p in m.parameters():
        p.requires_grad_(False)
    T = int(kw["sample_rate"] * secs)
    x = (2 * torch.rand(B, 1, T, device=dev) - 1) if dist == "uniform" else torch.randn(B, 1, T, device=dev)
    if bf16:
        x = x.to(torch.bfloat16)
    with torch.no_grad():
        out = m(x)
        med, p10, p90 = timed_calls(lambda: m(x))
    frames = out.shape[0] * out.shape[2]
    K, hop, F = m._complex_conv._kernel_size, m._pooling.strides, kw["n_filters"]
    flops = (2 * 2 * F * K * hop + 2 * F * K) * frames
    io_bytes = x.numel() * x.element_size() + out.numel() * out.element_size()
    algo_id = _native.load().leaf_auto_algo(B, T, F, K, hop)
    algo = ALGO_NAMES.get(algo_id, "?")
    # fp32 flops the selected kernel executes per call (bench.executed_flops mirrors the kernels' plans) over the WHOLE
    # forward's median time (tables + main kernel + finalize): a lower bound of the main kernel's own fraction
    executed, _ = bench.executed_flops(algo_id, m._complex_conv._kernel.detach(), m._pooling.weights.detach(), B, T, F, K, hop, _native.load(),
                                       m._pooling._bias.detach())
    print(json.dumps({"config": name, "in": list(x.shape), "in_dtype": str(x.dtype).replace("torch.", ""),
                      "out": list(out.shape), "out_dtype": str(out.dtype).replace("torch.", ""), "algo": algo,
                      "ms_median": round(med, 4), "ms_p10": round(p10, 4), "ms_p90": round(p90, 4),
                      "frames_per_s": round(frames / med * 1e3),
                      "algorithmic_TFLOPs": round(flops / med / 1e9, 1),
                      "executed_GFLOP_per_call": round(executed / 1e9, 2),
                      "executed_TFLOPs": round(executed / med / 1e9, 1),
                      "frac_of_fp32_valu_peak": round(executed / med / 1e9 / bench.PEAK_FP32_VALU_TFLOPS, 4),
                      "algorithmic_GBps": round(io_bytes / med / 1e6, 1)}), flush=True)
