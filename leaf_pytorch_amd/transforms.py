"""On-device counterparts of the reference's waveform transforms that sit directly in front of the frontend
(utilities/data/raw_transforms.py): the crops are views, PeakNormalization is a HIP kernel.  They take batched device
tensors ``(B, T)`` or ``(B, 1, T)`` (the reference versions run per clip on the CPU inside DataLoader workers)."""
import random

import torch

from . import _native


class RandomCrop:
    """raw_transforms.py:121-127 -- a random window of ``size`` samples (one offset per call, shared by the batch)."""

    def __init__(self, size: int):
        self.size = size

    def __call__(self, signal: torch.Tensor) -> torch.Tensor:
        start = random.randint(0, signal.shape[-1] - self.size)
        return signal[..., start: start + self.size]


class CenterCrop:
    """raw_transforms.py:130-140 -- the central ``size`` samples; shorter inputs pass unchanged."""

    def __init__(self, size: int):
        self.size = size

    def __call__(self, signal: torch.Tensor) -> torch.Tensor:
        if signal.shape[-1] > self.size:
            start = (signal.shape[-1] - self.size) // 2
            return signal[..., start: start + self.size]
        return signal


class PeakNormalization:
    """raw_transforms.py:334-345 -- ``torch_audiomentations.PeakNormalization(apply_to="only_too_loud_sounds", p=1)``:
    every clip whose peak |x| exceeds 1 is divided by its peak, quieter clips are returned unchanged.  (The third-party
    package is not in this image: its documented behaviour is restated, parity unpinned.)"""

    def __init__(self, sr: int = 16000):
        self.sr = sr                      # kept for signature compatibility; the operation does not depend on it

    def __call__(self, x: torch.Tensor) -> torch.Tensor:
        return _native.peak_normalize(x)
