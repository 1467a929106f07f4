"""Constructor-time initial values for the Gabor filterbank parameter (runs once, on the host).

Mirrors the interface of the reference's ``leaf_pytorch/initializers.py:7-24`` (``GaborInit``) and
``leaf_pytorch/filters.py:9-58`` (``GaborFilter``) so that ``Leaf(initializer="default")`` works
without torchaudio.  The reference gets its mel filterbank from
``torchaudio.functional.melscale_fbanks`` (filters.py:48-54; un-vendored third-party dependency,
pinned only as ``torchaudio >= 0.9.0``); ``htk_mel_filterbank`` below restates that function's
published HTK-scale triangular construction (norm=None).  Parity of this initial value with the
reference is UNPINNED (no torchaudio in the build image); it only seeds ``_complex_conv._kernel``,
which every checkpoint overrides.
"""
from __future__ import annotations

import math

import torch


def htk_mel_filterbank(n_freqs: int, f_min: float, f_max: float, n_mels: int, sample_rate: int) -> torch.Tensor:
    """(n_freqs, n_mels) triangular filters, HTK mel scale 2595*log10(1+f/700), no area normalisation."""
    freqs = torch.linspace(0, sample_rate // 2, n_freqs)
    mel_lo = 2595.0 * math.log10(1.0 + f_min / 700.0)
    mel_hi = 2595.0 * math.log10(1.0 + f_max / 700.0)
    edges = 700.0 * (10.0 ** (torch.linspace(mel_lo, mel_hi, n_mels + 2) / 2595.0) - 1.0)
    width = edges[1:] - edges[:-1]
    dist = edges.unsqueeze(0) - freqs.unsqueeze(1)                  # (n_freqs, n_mels + 2)
    rising = -dist[:, :-2] / width[:-1]
    falling = dist[:, 2:] / width[1:]
    return torch.minimum(rising, falling).clamp_min(0.0)


class GaborFilter:
    """filters.py:9-58 -- mel filterbank -> (centre frequency, width) pairs of matching Gabor filters."""

    def __init__(self, n_filters: int = 40, min_freq: float = 0.0, max_freq: float = 8000.0,
                 sample_rate: int = 16000, window_len: int = 401, n_fft: int = 512,
                 normalize_energy: bool = False):
        self.n_filters, self.min_freq, self.max_freq = n_filters, min_freq, max_freq
        self.sample_rate, self.window_len, self.n_fft = sample_rate, window_len, n_fft
        self.normalize_energy = normalize_energy

    def mel_filters(self) -> torch.Tensor:
        bank = htk_mel_filterbank(self.n_fft // 2 + 1, self.min_freq, self.max_freq, self.n_filters,
                                  self.sample_rate).t()
        if self.normalize_energy:
            peak = bank.max(dim=1, keepdim=True).values
            area = peak * ((bank > 0).float().sum(dim=1, keepdim=True) + 2) * math.pi / self.n_fft
            bank = bank / area
        return bank

    def gabor_params_from_mels(self) -> torch.Tensor:
        amp = self.mel_filters().sqrt()
        peak = amp.max(dim=1, keepdim=True).values
        centre_bin = amp.argmax(dim=1)
        fwhm_bins = (amp >= peak / 2.0).float().sum(dim=1)
        scale = torch.sqrt(2.0 * torch.log(torch.tensor(2.0))) * self.n_fft
        mu = centre_bin * 2 * math.pi / self.n_fft
        sigma = scale / (math.pi * fwhm_bins)
        return torch.stack([mu, sigma], dim=1).float()


class GaborInit:
    """initializers.py:7-24 -- callable ``(shape) -> Tensor(F,2)`` used as GaborConv1d's initializer."""

    def __init__(self, default_window_len: int = 401, **kwargs):
        self.def_win_len = default_window_len
        self._kwargs = kwargs

    def __call__(self, shape, dtype=None) -> torch.Tensor:
        if len(shape) != 2:
            raise NotImplementedError("implementation incomplete. Use even valued filter dimensions")
        return GaborFilter(n_filters=shape[0], window_len=self.def_win_len, **self._kwargs).gabor_params_from_mels()
