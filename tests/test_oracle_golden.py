"""Pin the CPU oracle (oracle/leaf_oracle.py) to the reference via the committed golden vectors."""
import math
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

from conftest import GOLDEN_DIR, Golden, golden_names, rel_err
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


REFERENCE = os.environ.get("LEAF_REFERENCE", "/root/reference")


@pytest.mark.skipif(not os.path.isdir(os.path.join(REFERENCE, "leaf_pytorch")),
                    reason="the reference tree only exists in the build container")
def test_golden_generator_reproduces_committed_fixtures():
    """The committed recipe must still run from a clean checkout and give the committed bytes: regenerates all
    fixtures from the imported reference into a temp dir (tests/golden/make_golden.py --check) -- a fresh
    interpreter, because this process already has the repo's own ``leaf_pytorch`` shim importable."""
    env = dict(os.environ, PYTHONDONTWRITEBYTECODE="1", LEAF_REFERENCE=REFERENCE)
    env.pop("PYTHONPATH", None)
    r = subprocess.run([sys.executable, "-B", os.path.join(GOLDEN_DIR, "make_golden.py"), "--check"],
                       capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "all bit-identical" in r.stdout


@pytest.mark.parametrize("n_filters,sample_rate", [(40, 16000), (80, 32000), (64, 16000), (40, 22050)])
def test_mel_init_matches_an_independent_htk_filterbank(n_filters, sample_rate):
    """filters.py:28-58 on top of an INDEPENDENT implementation of the HTK triangular filterbank that is in the
    image (transformers.audio_utils.mel_filter_bank, norm=None, mel_scale="htk") -- not torchaudio, so the default
    initial kernel stays labelled parity-unpinned; but two unrelated restatements of torchaudio's documented
    construction agree bit for bit on (argmax bin, FWHM count) for every shipped (filters, sample rate) pair."""
    au = pytest.importorskip("transformers.audio_utils")
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")                      # "a mel filter has all zero values" at 80 filters / 257 bins
        bank = au.mel_filter_bank(257, n_filters, 60.0, 7800.0, sample_rate, norm=None, mel_scale="htk")
    amp = torch.from_numpy(np.asarray(bank)).float().t().sqrt()           # (F, 257): filters.py:35
    peak = amp.max(dim=1, keepdim=True).values
    mu = amp.argmax(dim=1) * 2 * math.pi / 512                             # filters.py:36-40
    fwhm = (amp >= peak / 2).float().sum(dim=1)
    sigma = torch.sqrt(2.0 * torch.log(torch.tensor(2.0))) * 512 / (math.pi * fwhm)
    independent = torch.stack([mu, sigma], dim=1).float()
    assert torch.equal(lo.mel_gabor_init(n_filters, sample_rate), independent)
    from leaf_pytorch_amd.initializers import GaborInit
    product = GaborInit(default_window_len=int(sample_rate * 25 // 1000 + 1), sample_rate=sample_rate,
                        min_freq=60.0, max_freq=7800.0)((n_filters, 2))
    assert torch.equal(product, independent)
