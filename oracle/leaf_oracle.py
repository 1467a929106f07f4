"""CPU oracle for the LEAF frontend forward path -- TEST INFRASTRUCTURE ONLY.

This file is a from-scratch restatement, in stock torch CPU ops, of the arithmetic that
``leaf_pytorch.frontend.Leaf.forward`` (reference ``leaf_pytorch/frontend.py:78-89``) performs.
It exists so that the hand-written HIP kernels in ``leaf_pytorch_amd/csrc`` can be checked
against something that is itself pinned to the reference (``tests/golden/*.npz``, produced by
``tests/golden/make_golden.py`` from the imported reference).

Only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` may
import this module.  The product package ``leaf_pytorch_amd`` never imports it: there is no CPU
fallback in the product path.

Parity status
-------------
* forward arithmetic (stages a2..a10 of SURVEY.md section 8a): PINNED by golden vectors made from
  the reference itself (see tests/test_oracle_golden.py).
* ``mel_gabor_init`` (default initial value of ``_complex_conv._kernel``): PARITY UNPINNED.  The
  reference obtains the mel filterbank from ``torchaudio.functional.melscale_fbanks``
  (``leaf_pytorch/filters.py:48-54``), a third-party dependency that is neither vendored under
  /root/reference nor installed in this image (README pins only ``torchaudio >= 0.9.0``).  The
  function below restates torchaudio's published HTK triangular filterbank; it only decides the
  *initial* parameter values, which every checkpoint / state_dict overrides.

Every function cites the reference lines it follows.  ``dtype`` selects float32 (same arithmetic
type as the reference) or float64 (a higher-precision yardstick for error budgets).
"""
from __future__ import annotations

import math
from dataclasses import dataclass
from typing import Dict, Optional, Tuple

import torch
import torch.nn.functional as F

__all__ = [
    "LeafGeometry", "geometry", "same_padding", "constrain_gabor", "gabor_taps",
    "gabor_filterbank", "squared_modulus", "lowpass_window", "gaussian_pool", "ema_scan",
    "pcen", "leaf_forward", "mel_gabor_init", "default_params",
]

FLOOR_POOLED = 1e-5   # frontend.py:84  torch.maximum(outputs, 1e-5)
PCEN_FLOOR = 1e-12    # frontend.py:70  PCENLayer(floor=1e-12)


@dataclass(frozen=True)
class LeafGeometry:
    n_filters: int
    sample_rate: int
    window_size: int   # K
    hop: int
    pad_left: int
    pad_right: int

    def n_frames(self, n_samples: int) -> int:
        # strided conv over a "same"-padded signal: floor((T + padL + padR - K) / hop) + 1
        return (n_samples + self.pad_left + self.pad_right - self.window_size) // self.hop + 1


def same_padding(kernel_size: int) -> Tuple[int, int]:
    """utils.py:5-10 -- TF-style "same" padding: (k//2 + k%2 - 1, k//2)."""
    return kernel_size // 2 + kernel_size % 2 - 1, kernel_size // 2


def geometry(n_filters: int = 40, sample_rate: int = 16000, window_len: float = 25.0,
             window_stride: float = 10.0) -> LeafGeometry:
    """frontend.py:38-39 -- window/hop in samples from milliseconds."""
    k = int(sample_rate * window_len // 1000 + 1)
    hop = int(sample_rate * window_stride // 1000)
    pl, pr = same_padding(k)
    return LeafGeometry(n_filters, sample_rate, k, hop, pl, pr)


def constrain_gabor(kernel: torch.Tensor, kernel_size: int) -> torch.Tensor:
    """convolution.py:15-22 -- clamp centre frequency to [0, pi] and width to [4c, K c]."""
    c = math.sqrt(2.0 * math.log(2.0)) / math.pi
    if kernel.dtype == torch.float32:
        # the reference builds the bounds from float32 tensors
        c32 = (torch.sqrt(2.0 * torch.log(torch.tensor(2.0))) / math.pi)
        lo, hi = float(4 * c32), float(kernel_size * c32)
    else:
        lo, hi = 4 * c, kernel_size * c
    mu = kernel[:, 0].clamp(0.0, math.pi)
    sigma = kernel[:, 1].clamp(lo, hi)
    return torch.stack([mu, sigma], dim=1)


def gabor_taps(kernel: torch.Tensor, kernel_size: int) -> Tuple[torch.Tensor, torch.Tensor]:
    """impulse_responses.py:5-16, 66-71 -- complex Gabor impulse responses.

    ``kernel`` is the *constrained* (F,2) parameter.  Returns (real, imag), each (F,K):
        h_f[t] = exp(-t^2 / (2 sigma_f^2)) / (sqrt(2 pi) sigma_f) * exp(i mu_f t),
        t = -(K//2) .. (K+1)//2 - 1
    """
    dt = kernel.dtype
    t = torch.arange(-(kernel_size // 2), (kernel_size + 1) // 2, dtype=dt)
    mu, sigma = kernel[:, 0:1], kernel[:, 1:2]
    norm = 1.0 / (math.sqrt(2.0 * math.pi) * sigma)
    envelope = torch.exp((1.0 / (2.0 * sigma * sigma)) * (-(t * t)).unsqueeze(0))
    phase = mu * t.unsqueeze(0)
    return norm * torch.cos(phase) * envelope, norm * torch.sin(phase) * envelope


def gabor_filterbank(x: torch.Tensor, taps_re: torch.Tensor, taps_im: torch.Tensor) -> torch.Tensor:
    """convolution.py:77-99 -- zero "same" pad, cross-correlate with 2F interleaved rows.

    x: (B,1,T).  Output (B,2F,T): row 2f = Re, row 2f+1 = Im.
    """
    n_f, k = taps_re.shape
    bank = torch.stack([taps_re, taps_im], dim=1).reshape(2 * n_f, 1, k)
    return F.conv1d(F.pad(x, same_padding(k)), bank)


def squared_modulus(y: torch.Tensor) -> torch.Tensor:
    """frontend.py:15-19 -- re^2 + im^2 over interleaved channel pairs: (B,2F,T) -> (B,F,T).

    (The reference writes it as 2 * avgpool_2(y^2) over the channel axis.)
    """
    b, c2, t = y.shape
    sq = (y * y).reshape(b, c2 // 2, 2, t)
    return 2.0 * (0.5 * (sq[:, :, 0] + sq[:, :, 1]))


def lowpass_window(width: torch.Tensor, kernel_size: int) -> torch.Tensor:
    """impulse_responses.py:74-80 -- un-normalised Gaussian window per channel: (F,) -> (F,K)."""
    s = width.reshape(-1, 1).clamp(2.0 / kernel_size, 0.5)
    j = torch.arange(kernel_size, dtype=width.dtype).unsqueeze(0)
    half = 0.5 * (kernel_size - 1)
    return torch.exp(-0.5 * ((j - half) / (s * half)) ** 2)


def gaussian_pool(e: torch.Tensor, window: torch.Tensor, bias: Optional[torch.Tensor], hop: int) -> torch.Tensor:
    """pooling.py:31-42 -- depth-wise strided cross-correlation (+bias) over the same-padded energy."""
    n_f, k = window.shape
    return F.conv1d(F.pad(e, same_padding(k)), window.unsqueeze(1), bias=bias, stride=hop, groups=n_f)


def ema_scan(p: torch.Tensor, coeff: torch.Tensor) -> torch.Tensor:
    """postprocessing.py:13-28 -- per-channel first-order EMA along time, state initialised to p[...,0]."""
    w = coeff.clamp(0.0, 1.0).reshape(1, -1)
    state = p[:, :, 0]
    out = torch.empty_like(p)
    for i in range(p.shape[-1]):
        state = w * p[:, :, i] + (1.0 - w) * state
        out[:, :, i] = state
    return out


def pcen(p: torch.Tensor, alpha: torch.Tensor, delta: torch.Tensor, root: torch.Tensor,
         ema_coeff: torch.Tensor, return_ema: bool = False):
    """postprocessing.py:62-69 -- (p / (floor + M)^alpha + delta)^(1/r) - delta^(1/r)."""
    a = alpha.clamp(max=1.0).reshape(1, -1, 1)
    inv_r = (1.0 / root.clamp(min=1.0)).reshape(1, -1, 1)
    d = delta.reshape(1, -1, 1)
    m = ema_scan(p, ema_coeff)
    out = (p / (PCEN_FLOOR + m) ** a + d) ** inv_r - d ** inv_r
    return (out, m) if return_ema else out


def leaf_forward(x: torch.Tensor, params: Dict[str, torch.Tensor], geo: LeafGeometry,
                 pcen_compression: bool = True, dtype: torch.dtype = torch.float32,
                 return_stages: bool = False):
    """frontend.py:78-89 -- the whole forward.  ``params`` uses the reference's state_dict keys."""
    x = x.to(dtype)
    p = {k: v.to(dtype) for k, v in params.items()}     # keeps the autograd graph: tests differentiate through the oracle
    kern = constrain_gabor(p["_complex_conv._kernel"], geo.window_size)
    hr, hi = gabor_taps(kern, geo.window_size)
    y = gabor_filterbank(x, hr, hi)
    e = squared_modulus(y)
    g = lowpass_window(p["_pooling.weights"].reshape(-1), geo.window_size)
    pooled = gaussian_pool(e, g, p["_pooling._bias"], geo.hop)
    pooled = torch.clamp(pooled, min=FLOOR_POOLED)
    stages = {"taps_re": hr, "taps_im": hi, "energy": e, "lowpass": g, "pooled": pooled}
    out = pooled
    if pcen_compression:
        out, m = pcen(pooled, p["_compression.alpha"], p["_compression.delta"], p["_compression.root"],
                      p["_compression.ema._weights"], return_ema=True)
        stages["ema"] = m
    return (out, stages) if return_stages else out


# --------------------------------------------------------------------------------------------
# parameter initialisation (ctor-time only)
# --------------------------------------------------------------------------------------------

def _hz_to_mel_htk(f):
    return 2595.0 * math.log10(1.0 + f / 700.0)


def mel_gabor_init(n_filters: int = 40, sample_rate: int = 16000, min_freq: float = 60.0,
                   max_freq: float = 7800.0, n_fft: int = 512) -> torch.Tensor:
    """initializers.py:13-18 + filters.py:28-58 -- (F,2) Gabor (mu, sigma) from a mel filterbank.

    PARITY UNPINNED (see module docstring): the triangular HTK filterbank of
    torchaudio.functional.melscale_fbanks(n_freqs=n_fft//2+1, f_min, f_max, n_mels, sample_rate,
    norm=None, mel_scale="htk") is restated here from its published definition.
    """
    n_freqs = n_fft // 2 + 1
    all_freqs = torch.linspace(0, sample_rate // 2, n_freqs, dtype=torch.float32)
    m_pts = torch.linspace(_hz_to_mel_htk(min_freq), _hz_to_mel_htk(max_freq), n_filters + 2,
                           dtype=torch.float32)
    f_pts = 700.0 * (10.0 ** (m_pts / 2595.0) - 1.0)
    f_diff = f_pts[1:] - f_pts[:-1]
    slopes = f_pts.unsqueeze(0) - all_freqs.unsqueeze(1)          # (n_freqs, n_filters+2)
    down = -slopes[:, :-2] / f_diff[:-1]
    up = slopes[:, 2:] / f_diff[1:]
    fb = torch.clamp(torch.minimum(down, up), min=0.0).t()        # (n_filters, n_freqs)
    root = torch.sqrt(fb)                                         # filters.py:30
    centre = torch.argmax(root, dim=1)                            # filters.py:31
    peak = root.max(dim=1, keepdim=True).values
    fwhm = (root >= peak / 2.0).float().sum(dim=1)                # filters.py:34
    coeff = torch.sqrt(2.0 * torch.log(torch.tensor(2.0))) * n_fft
    return torch.stack([centre * 2 * math.pi / n_fft, coeff / (math.pi * fwhm)], dim=1).float()


def default_params(geo: LeafGeometry, pcen_compression: bool = True,
                   kernel: Optional[torch.Tensor] = None) -> Dict[str, torch.Tensor]:
    """Initial parameter values of Leaf(): convolution.py:58, pooling.py:18-22, postprocessing.py:52-54,11."""
    n_f = geo.n_filters
    out = {
        "_complex_conv._kernel": kernel if kernel is not None else mel_gabor_init(n_f, geo.sample_rate),
        "_pooling.weights": torch.full((1, 1, n_f, 1), 0.4),
        "_pooling._bias": torch.ones(n_f),
    }
    if pcen_compression:
        out.update({
            "_compression.alpha": torch.full((n_f,), 0.96),
            "_compression.delta": torch.full((n_f,), 2.0),
            "_compression.root": torch.full((n_f,), 2.0),
            "_compression.ema._weights": torch.full((n_f,), 0.04),
        })
    return out


def peak_normalize(x: torch.Tensor) -> torch.Tensor:
    """utilities/data/raw_transforms.py:334-345: ``torch_audiomentations.PeakNormalization(apply_to=
    "only_too_loud_sounds", p=1.)`` -- per clip, divide by the peak |x| when it exceeds 1.  torch_audiomentations is a
    third-party dependency absent from this image (requirements: unpinned): its documented behaviour is restated,
    **parity unpinned**."""
    flat = x.reshape(x.shape[0], -1)
    peak = flat.abs().amax(dim=1, keepdim=True)
    scaled = torch.where(peak > 1.0, flat / peak.clamp_min(1e-30), flat)
    return scaled.reshape(x.shape)
