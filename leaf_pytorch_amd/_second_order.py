"""Gradients of gradients through ``Leaf`` (``create_graph=True``: gradient penalties, Hessian-vector products, MAML-style loops).

The reference's forward is a graph of stock differentiable ops (frontend.py:78-89 over convolution.py:71-99, pooling.py:31-42,
postprocessing.py:13-28, 62-69), so autograd can differentiate its backward again.  Here the forward and the FIRST-order backward are
hand-written HIP kernels (``leaf_amd::forward_train`` / ``leaf_amd::backward``); this module gives ``leaf_amd::backward`` its own
autograd formula, so that the second order exists too.  It is reached from nowhere else: ``Leaf.forward`` and the first-order
backward never run this code.

How: G = ``leaf_amd::backward``(x, theta, grad_out) is J(x, theta)^T grad_out.  For cotangents v on G, the formula needs
d<v, G>/d(x, theta, grad_out).  It rebuilds the forward ON THE SAME DEVICE from differentiable torch ops (``composite_forward`` below:
the filterbank as a product of rocFFT spectra, the pooling as a strided window sum, the EMA as the recurrence it is), takes G from it
with ``create_graph=True`` and differentiates <v, G>.  Second order only: the result carries no graph of its own (a third order
would differentiate the composite directly).
Cost: that of the reference's own double backward on this device (seconds at training batch sizes, not the fused kernels' fraction
of a millisecond) -- second-order training loops are rare for a frontend, and there is no hand-written kernel for them.

The tensors are those of ``leaf_amd::backward``, which accepts HIP tensors only: there is no CPU path in the product
(tests/test_host_logic.py runs ``composite_forward`` and the formula on the CPU only to pin them against the oracle).
"""
from __future__ import annotations

import math

import torch

FLOOR_POOLED = 1e-5        # frontend.py:86: torch.maximum(outputs, 1e-5)
PCEN_FLOOR = 1e-12         # frontend.py:65-69: the floor Leaf constructs its PCENLayer with


def composite_forward(x, kernel, pool_w, pool_b, alpha, delta, root, ema_w, K: int, hop: int) -> torch.Tensor:
    """frontend.py:78-89 from differentiable torch ops, in the dtype of ``x``.  x: (B, 1, T); returns (B, F, T')."""
    B, _, T = x.shape
    F = kernel.shape[0]
    dt = x.dtype
    pad_l, pad_r = K // 2 + K % 2 - 1, K // 2                                   # utils.py:5-10
    # convolution.py:15-22: the constraint; its bounds come from float32 tensors in the reference
    c32 = float(torch.sqrt(2.0 * torch.log(torch.tensor(2.0))) / math.pi) if dt == torch.float32 else math.sqrt(2.0 * math.log(2.0)) / math.pi
    mu = kernel[:, 0:1].clamp(0.0, math.pi)
    sigma = kernel[:, 1:2].clamp(4 * c32, K * c32)
    # impulse_responses.py:5-16, 66-71
    t = torch.arange(-(K // 2), (K + 1) // 2, dtype=dt, device=x.device).unsqueeze(0)
    env = torch.exp(-(t * t) / (2.0 * sigma * sigma)) / (math.sqrt(2.0 * math.pi) * sigma)
    taps = torch.complex(env * torch.cos(mu * t), env * torch.sin(mu * t))      # (F, K)
    # convolution.py:77-99: y[n] = sum_k h[k] xz[n + k] (cross-correlation over the same-padded clip) as a product of spectra
    n = 1 << (T + 2 * K).bit_length()
    xz = torch.nn.functional.pad(x[:, 0], (pad_l, pad_r))                       # (B, T + K - 1)
    X = torch.fft.fft(xz.to(taps.dtype), n=n)                                   # (B, n)
    Hc = torch.fft.fft(taps.conj(), n=n).conj()                                 # correlation: conj(FFT(conj(h)))
    y = torch.fft.ifft(X.unsqueeze(1) * Hc.unsqueeze(0))[..., :T]               # (B, F, T) complex: Re / Im rows of the reference
    e = y.real * y.real + y.imag * y.imag                                       # frontend.py:15-19
    # impulse_responses.py:74-80 + pooling.py:31-42
    s = pool_w.reshape(-1, 1).clamp(2.0 / K, 0.5)
    j = torch.arange(K, dtype=dt, device=x.device).unsqueeze(0)
    half = 0.5 * (K - 1)
    g = torch.exp(-0.5 * ((j - half) / (s * half)) ** 2)                        # (F, K)
    frames = torch.nn.functional.pad(e, (pad_l, pad_r)).unfold(-1, K, hop)      # (B, F, T', K)
    pooled = (frames * g.unsqueeze(0).unsqueeze(2)).sum(-1) + pool_b.reshape(1, -1, 1)
    pooled = pooled.clamp(min=FLOOR_POOLED)
    if alpha is None:
        return pooled
    # postprocessing.py:13-28 (state starts at the first frame) and 62-69
    w = ema_w.clamp(0.0, 1.0).reshape(1, -1)
    state = pooled[:, :, 0]
    ms = []
    for i in range(pooled.shape[-1]):
        state = w * pooled[:, :, i] + (1.0 - w) * state
        ms.append(state)
    m = torch.stack(ms, dim=-1)
    a = alpha.clamp(max=1.0).reshape(1, -1, 1)
    inv_r = (1.0 / root.clamp(min=1.0)).reshape(1, -1, 1)
    d = delta.reshape(1, -1, 1)
    return (pooled / (PCEN_FLOOR + m) ** a + d) ** inv_r - d ** inv_r


def setup_context(ctx, inputs, output):
    x, kernel, pool_w, pool_b, alpha, delta, root, ema_w, K, hop, grad_out, pooled_raw, need_dx, flags = inputs
    ctx.pcen = alpha is not None
    ctx.geom = (K, hop)
    ctx.need_dx = bool(need_dx)
    ctx.save_for_backward(x, kernel, pool_w, pool_b, grad_out, *([alpha, delta, root, ema_w] if ctx.pcen else []))


def backward(ctx, grads):
    """Cotangents ``grads`` on (g_kernel, g_pool_w, g_pool_b, g_alpha, g_delta, g_root, g_ema_w, g_x) -> gradients on the op's inputs."""
    K, hop = ctx.geom
    saved = ctx.saved_tensors
    x, kernel, pool_w, pool_b, grad_out = saved[:5]
    with torch.enable_grad():
        leaves = [t.detach().requires_grad_(True) for t in (x, kernel, pool_w, pool_b, *saved[5:])]
        go = grad_out.detach().requires_grad_(True)
        pc = leaves[4:] if ctx.pcen else [None] * 4
        out = composite_forward(leaves[0], leaves[1], leaves[2], leaves[3], *pc, K, hop)
        wrt = leaves[1:] + ([leaves[0]] if ctx.need_dx else [])                 # the op's output order: parameters, then x
        vs = [grads[i] for i in range(3)] + ([grads[i] for i in range(3, 7)] if ctx.pcen else []) + ([grads[7]] if ctx.need_dx else [])
        G = torch.autograd.grad(out, wrt, go, create_graph=True, allow_unused=True)
        s = None
        for v, g in zip(vs, G):
            if v is None or g is None:
                continue
            term = (v.detach().reshape(g.shape) * g).sum()
            s = term if s is None else s + term
        if s is None:
            res = [None] * (len(leaves) + 1)
        else:
            res = torch.autograd.grad(s, [*leaves, go], allow_unused=True)
    gx, gk, gpw, gpb = res[0], res[1], res[2], res[3]
    gpc = list(res[4:8]) if ctx.pcen else [None] * 4
    ggo = res[-1]
    # inputs: x, kernel, pool_w, pool_b, alpha, delta, root, ema_w, K, hop, grad_out, pooled_raw, need_dx, flags
    return gx, gk, gpw, gpb, *gpc, None, None, ggo, None, None, None


def register() -> None:
    torch.library.register_autograd("leaf_amd::backward", backward, setup_context=setup_context)
