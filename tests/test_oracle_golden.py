"""Pin the CPU oracle (oracle/leaf_oracle.py) to the reference via the committed golden vectors."""
import math

import numpy as np
import pytest
import torch

from conftest import Golden, golden_names, rel_err
from oracle import leaf_oracle as lo

# The reference's own fp32 noise floor vs fp64 is 2e-6 rel; the oracle runs the same
# op sequence in the same precision, so it must agree far below the 1e-4 north-star tolerance.
TOL_FP32 = 2e-6
TOL_FP64_VS_REF32 = 2e-5


def test_fixture_inventory():
    names = golden_names()
    assert len(names) >= 17
    for must in ("default_b2", "clamps_b2", "pcen_off_b2", "even_k_22k_b1", "f80_32k_b1", "len_1_b1"):
        assert must in names


def test_forward_fp32_matches_reference(golden):
    out, st = lo.leaf_forward(golden.x, golden.params, golden.geometry(), golden.pcen, torch.float32, True)
    ref = golden["out"]
    assert out.shape == ref.shape
    assert rel_err(out, ref) < TOL_FP32, golden.name
    assert rel_err(st["pooled"], golden["pooled"]) < TOL_FP32
    if golden.pcen:
        assert rel_err(st["ema"], golden["ema"]) < TOL_FP32


def test_forward_fp64_matches_reference(golden):
    out = lo.leaf_forward(golden.x, golden.params, golden.geometry(), golden.pcen, torch.float64)
    assert rel_err(out, golden["out"]) < TOL_FP64_VS_REF32, golden.name


@pytest.mark.parametrize("name", ["default_b2", "clamps_b2", "legacy_complex_b1", "even_k_22k_b1"])
def test_taps_match_reference(name):
    g = Golden(name)
    kern = lo.constrain_gabor(g.params["_complex_conv._kernel"], g.window_size)
    hr, hi = lo.gabor_taps(kern, g.window_size)
    bank = torch.stack([hr, hi], dim=1).reshape(2 * g.n_filters, g.window_size)
    ref = g["taps"]
    assert bank.shape == ref.shape
    assert float((bank - ref).abs().max()) < 2e-8          # max |h| ~ 0.27 at the sigma clamp; ~1 ulp


def test_energy_windows_match_reference(golden):
    _, st = lo.leaf_forward(golden.x, golden.params, golden.geometry(), golden.pcen, torch.float32, True)
    i = 0
    while golden.has(f"energy_{i}"):
        a, b = (int(v) for v in golden[f"energy_{i}_range"])
        ref = golden[f"energy_{i}"]
        got = st["energy"][:, :, a:b]
        scale = float(ref.abs().max()) + 1e-30
        assert float((got - ref).abs().max()) / scale < 5e-6
        i += 1


def test_frame_count_formula():
    geo = lo.geometry()
    for t, frames in ((1, 1), (159, 1), (160, 1), (161, 2), (15999, 100), (16000, 100), (16001, 101)):
        assert geo.n_frames(t) == frames
        assert geo.n_frames(t) == (t - 1) // geo.hop + 1          # SURVEY 3.1 closed form
    assert lo.same_padding(401) == (200, 200) and lo.same_padding(552) == (275, 276)
    assert (lo.geometry(40, 22050).window_size, lo.geometry(40, 22050).hop) == (552, 220)
    assert (lo.geometry(80, 32000).window_size, lo.geometry(80, 32000).hop) == (801, 320)


def test_mel_init_matches_recorded_values():
    """Parity-unpinned initial kernel: check only the anchors SURVEY section 8c recorded from its probe."""
    k = lo.mel_gabor_init(40, 16000)
    import os
    from conftest import GOLDEN_DIR
    rec = torch.from_numpy(np.load(os.path.join(GOLDEN_DIR, "default_kernel_f40_16k.npz"))["kernel"])
    assert torch.equal(rec, k)
    assert k.shape == (40, 2)
    assert abs(float(k[0, 0]) - 0.036816) < 1e-5 and abs(float(k[0, 1]) - 95.944) < 2e-2
    assert abs(float(k[-1, 0]) - 2.8716) < 1e-3 and abs(float(k[-1, 1]) - 8.7222) < 1e-3
    bins = k[:, 0] * 512 / (2 * math.pi)
    assert float((bins - bins.round()).abs().max()) < 1e-4          # mu is an integer FFT bin
