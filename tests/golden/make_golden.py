#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ from the REFERENCE implementation.

Run in the build container only (the reference does not exist on the GPU box):

    python tests/golden/make_golden.py            # writes tests/golden/*.npz
    python tests/golden/make_golden.py --check    # regenerates into a temp dir, compares every array bit for bit
                                                  # with the committed fixtures (tests/test_oracle_golden.py runs this
                                                  # whenever /root/reference exists)

What it does: imports ``leaf_pytorch.frontend.Leaf`` from /root/reference (read-only, no bytecode
written), runs it on seeded inputs with EXPLICIT parameter tensors, and stores inputs, parameters,
per-stage intermediates and outputs.  Fixtures are data only -- no reference source text.

torchaudio: ``leaf_pytorch/filters.py:4`` imports it at module top, and it is not installed here.
An EMPTY placeholder module is registered so the import statement succeeds; nothing in it is ever
called, because every Leaf below is built with an explicit callable initializer (the mel init that
would call ``torchaudio.functional.melscale_fbanks`` is never executed).  The default-initial
kernel stored in ``default_kernel_f40_16k.npz`` therefore comes from OUR restatement
(oracle.leaf_oracle.mel_gabor_init) and is labelled parity-unpinned.
"""
import argparse
import contextlib
import importlib.util
import io
import os
import sys
import tempfile
import types

import numpy as np
import torch

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
REF = os.environ.get("LEAF_REFERENCE", "/root/reference")
sys.modules.setdefault("torchaudio", types.ModuleType("torchaudio"))   # empty placeholder, never called
# The repo ships an import-path shim that is also called ``leaf_pytorch`` (the drop-in boundary), so the repo root must
# NOT be on sys.path ahead of the reference: the reference goes first, anything else of the repo's that may have been
# put there (pytest, PYTHONPATH) is dropped for the duration of the import, and the one repo function needed here is
# loaded by file path.
sys.path[:] = [REF] + [p for p in sys.path if os.path.realpath(p or os.getcwd()) != os.path.realpath(REPO)]
for _name in [n for n in sys.modules if n == "leaf_pytorch" or n.startswith("leaf_pytorch.")]:
    del sys.modules[_name]

from leaf_pytorch.frontend import Leaf as RefLeaf            # noqa: E402  (the reference)

assert os.path.realpath(sys.modules["leaf_pytorch"].__file__).startswith(os.path.realpath(REF) + os.sep), \
    f"leaf_pytorch resolved to {sys.modules['leaf_pytorch'].__file__}, not to the reference under {REF}"

_spec = importlib.util.spec_from_file_location("_leaf_oracle_for_goldens", os.path.join(REPO, "oracle", "leaf_oracle.py"))
_oracle = importlib.util.module_from_spec(_spec)
sys.modules[_spec.name] = _oracle                            # dataclasses look their module up while the body executes
_spec.loader.exec_module(_oracle)
mel_gabor_init = _oracle.mel_gabor_init                      # only for initial kernel values (parity unpinned)

OUT = HERE                                                   # where fixtures are written (--check: a temp dir)


def build_ref(kernel, n_filters, sample_rate, pcen=True, legacy=False, window_len=25.0, window_stride=10.0):
    with contextlib.redirect_stdout(io.StringIO()):
        m = RefLeaf(n_filters=n_filters, sample_rate=sample_rate, window_len=window_len,
                    window_stride=window_stride, pcen_compression=pcen, use_legacy_complex=legacy,
                    initializer=lambda shape: kernel.clone())
    return m.eval()


def perturb(sd, seed, scale=0.1):
    g = torch.Generator().manual_seed(seed)
    out = {}
    for k, v in sd.items():
        out[k] = v * (1.0 + scale * (2 * torch.rand(v.shape, generator=g) - 1))
    return out


def run_case(name, x, model, sd=None, energy_windows=((0, 64),), keep_taps=False):
    if sd is not None:
        model.load_state_dict(sd, strict=True)
    rec = {"x": x.numpy()}
    for k, v in model.state_dict().items():
        rec["param:" + k] = v.numpy().copy()
    with torch.no_grad():
        y = model._complex_conv(x)
        e = model._activation(y)
        pooled = torch.maximum(model._pooling(e), torch.tensor(1e-5))
        out = model(x)
        if keep_taps:
            # taps as the conv consumes them: rows 2f=re, 2f+1=im  (K identity trick: feed a unit impulse)
            k = model._complex_conv._kernel_size
            imp = torch.zeros(1, 1, 2 * k + 1)
            imp[0, 0, k] = 1.0
            resp = model._complex_conv(imp)[0]               # (2F, 2K+1); resp[c, k - t_j] = w[c, j]
            pad_l = k // 2 + k % 2 - 1
            taps = torch.stack([resp[:, k + pad_l - j] for j in range(k)], dim=1)
            rec["taps"] = taps.numpy()
        for i, (a, b) in enumerate(energy_windows):
            rec[f"energy_{i}"] = e[:, :, a:b].numpy().copy()
            rec[f"energy_{i}_range"] = np.array([a, b])
        rec["pooled"] = pooled.numpy()
        if model._compression is not None:
            rec["ema"] = model._compression.ema(pooled).numpy()
        rec["out"] = out.numpy()
    rec["meta"] = np.array([model._complex_conv._filters, model._complex_conv._kernel_size,
                            model._pooling.strides, int(model._compression is not None)])
    path = os.path.join(OUT, name + ".npz")
    np.savez_compressed(path, **rec)
    print(f"{name}: x{tuple(x.shape)} -> out{tuple(out.shape)}  {os.path.getsize(path)/1024:.0f} KB")


def main():
    torch.manual_seed(0)
    k40 = mel_gabor_init(40, 16000)
    np.savez_compressed(os.path.join(OUT, "default_kernel_f40_16k.npz"), kernel=k40.numpy(),
                        note=np.array("stub-free restatement of torchaudio htk mel init; parity unpinned"))
    g = torch.Generator().manual_seed(1234)

    # 1. default Leaf, N(0,1) input (reference test_leaf.py:8 distribution), B=2
    x = torch.randn(2, 1, 16000, generator=g)
    run_case("default_b2", x, build_ref(k40, 40, 16000),
             energy_windows=((0, 64), (7968, 8032), (15936, 16000)), keep_taps=True)

    # 2. +-10% perturbed parameters, U(-1,1) input (peak-normalised audio distribution), B=3
    x = 2 * torch.rand(3, 1, 16000, generator=g) - 1
    m = build_ref(k40, 40, 16000)
    run_case("perturbed_uniform_b3", x, m, sd=perturb(m.state_dict(), 7),
             energy_windows=((100, 164), (15000, 15064)))

    # 3. every clamp active somewhere
    m = build_ref(k40, 40, 16000)
    sd = {k: v.clone() for k, v in m.state_dict().items()}
    kern = sd["_complex_conv._kernel"]
    kern[0, 0] = -0.3; kern[1, 0] = 3.5; kern[2, 1] = 0.5; kern[3, 1] = 400.0      # mu<0, mu>pi, sigma low/high
    pw = sd["_pooling.weights"]; pw[0, 0, 4, 0] = 0.001; pw[0, 0, 5, 0] = 0.9; pw[0, 0, 6, 0] = -1.0
    sd["_pooling._bias"][7] = -50.0; sd["_pooling._bias"][8] = 0.0                   # floor 1e-5 active
    sd["_compression.alpha"][9] = 1.7; sd["_compression.alpha"][10] = 0.3
    sd["_compression.root"][11] = 0.4; sd["_compression.root"][12] = 3.0
    sd["_compression.ema._weights"][13] = -0.2; sd["_compression.ema._weights"][14] = 1.6
    sd["_compression.delta"][15] = 0.5; sd["_compression.delta"][16] = 10.0
    x = torch.randn(2, 1, 4000, generator=g)
    run_case("clamps_b2", x, m, sd=sd, energy_windows=((0, 64), (3936, 4000)), keep_taps=True)

    # 4. PCEN off (BASELINE config 3 shape, smaller batch)
    x = 2 * torch.rand(2, 1, 16000, generator=g) - 1
    run_case("pcen_off_b2", x, build_ref(k40, 40, 16000, pcen=False))

    # 5. legacy-complex tap synthesis (all shipped cfgs set use_legacy_complex: True)
    x = torch.randn(1, 1, 8000, generator=g)
    run_case("legacy_complex_b1", x, build_ref(k40, 40, 16000, legacy=True), keep_taps=True)

    # 6. even window: sr=22050 -> K=552, hop=220, pad (275,276)
    k22 = mel_gabor_init(40, 22050)
    x = torch.randn(1, 1, 5000, generator=g)
    run_case("even_k_22k_b1", x, build_ref(k22, 40, 22050), energy_windows=((0, 64), (4936, 5000)), keep_taps=True)

    # 7. 80 filters / 32 kHz (K=801, hop=320)
    k80 = mel_gabor_init(80, 32000)
    x = 2 * torch.rand(1, 1, 9600, generator=g) - 1
    run_case("f80_32k_b1", x, build_ref(k80, 80, 32000), energy_windows=((4000, 4064),))

    # 8. 64 filters (AudioSet cfgs), non-default window/stride: 20 ms / 5 ms -> K=321, hop=80 (4 overlapping frames)
    k64 = mel_gabor_init(64, 16000)
    x = torch.randn(2, 1, 3000, generator=g)
    run_case("f64_win20_hop5_b2", x, build_ref(k64, 64, 16000, window_len=20.0, window_stride=5.0))

    # 9. ragged / edge lengths
    for t in (1, 2, 159, 160, 161, 401, 15999, 16001):
        x = torch.randn(1, 1, t, generator=g)
        run_case(f"len_{t}_b1", x, build_ref(k40, 40, 16000), energy_windows=((0, min(t, 64)),))

    # 10. non-overlapping frames: window 5 ms, stride 10 ms -> K=81 < hop=160
    x = torch.randn(2, 1, 2000, generator=g)
    run_case("short_window_b2", x, build_ref(k40, 40, 16000, window_len=5.0, window_stride=10.0))

    # 11. the empty batch (last, so the seeded stream above is untouched): the reference returns (0, F, T') for B = 0
    run_case("empty_b0", torch.zeros(0, 1, 1600), build_ref(k40, 40, 16000), energy_windows=())


def check() -> int:
    """Regenerate every fixture into a temp dir and compare with the committed files, array by array, bit for bit."""
    global OUT
    with tempfile.TemporaryDirectory(prefix="leaf_golden_") as tmp:
        OUT = tmp
        with contextlib.redirect_stdout(io.StringIO()):
            main()
        OUT = HERE
        fresh = sorted(f for f in os.listdir(tmp) if f.endswith(".npz"))
        committed = sorted(f for f in os.listdir(HERE) if f.endswith(".npz"))
        bad = []
        if fresh != committed:
            bad.append(f"file sets differ: generated {fresh} vs committed {committed}")
        for f in (f for f in fresh if f in committed):
            a, b = np.load(os.path.join(tmp, f)), np.load(os.path.join(HERE, f))
            if sorted(a.files) != sorted(b.files):
                bad.append(f"{f}: keys differ")
                continue
            for k in a.files:
                if a[k].dtype != b[k].dtype or a[k].shape != b[k].shape or a[k].tobytes() != b[k].tobytes():
                    bad.append(f"{f}:{k} differs")
    for line in bad:
        print("MISMATCH", line)
    print(f"{len(fresh)} fixtures regenerated from {REF}: " + ("all bit-identical" if not bad else f"{len(bad)} mismatches"))
    return 1 if bad else 0


if __name__ == "__main__":
    ap = argparse.ArgumentParser(description=__doc__.split("\n")[0])
    ap.add_argument("--check", action="store_true", help="regenerate into a temp dir and compare bit for bit")
    if ap.parse_args().check:
        sys.exit(check())
    main()
