"""The reference's OWN caller on top of the product (SURVEY 8b "who calls it").

Container-only: `/root/reference` does not exist on the GPU box, so everything here skips there (like the regeneration of the
golden fixtures).  Nothing of the reference is copied: its `models.classifier.Classifier` (models/classifier.py:7-18) and its
shipped `cfgs/**/*.cfg` are imported / read where they lie, in FRESH interpreters (`-B`: no bytecode into the read-only tree)
whose `sys.path` puts this repo AHEAD of the reference, so `from leaf_pytorch import get_frontend` (models/classifier.py:3)
resolves to the import-path shim `leaf_pytorch/` of this repo and `models.*` to the reference.

What is pinned, for every shipped cfg:  `Classifier(cfg).features` IS `leaf_pytorch_amd.frontend.Leaf`; its constructor arguments
follow frontend_helper.py:7-54 (default_args / n_filters / initializer / use_legacy_complex); the checkpoint keys are the seven
(or three) `features.*` names of the reference; and a `model_state_dict` PRODUCED BY THE REFERENCE's Leaf inside the reference's
Classifier loads strictly into the product-backed Classifier and back (train.py:33-49's checkpoint format).  No forward runs
here (the product has no CPU path); the forward / backward of a Classifier-shaped caller is tests/test_gpu_dropin.py's."""
import glob
import json
import os
import subprocess
import sys
import tempfile

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REFERENCE = os.environ.get("LEAF_REFERENCE", "/root/reference")
CFGS = sorted(glob.glob(os.path.join(REFERENCE, "cfgs", "**", "*.cfg"), recursive=True))

pytestmark = pytest.mark.skipif(not os.path.isdir(os.path.join(REFERENCE, "models")),
                                reason="the reference tree only exists in the build container")

SEVEN = ["features._complex_conv._kernel", "features._pooling.weights", "features._pooling._bias", "features._compression.alpha",
         "features._compression.delta", "features._compression.root", "features._compression.ema._weights"]

# Runs in a fresh interpreter.  argv: mode ("product" | "reference"), repo, reference, json-out, state-dict dir, cfg paths...
CHILD = r'''
import contextlib, io, json, os, sys, types
mode, repo, ref, out_path, sd_dir = sys.argv[1:6]
cfgs = sys.argv[6:]
sys.dont_write_bytecode = True
sys.path[:] = ([repo, ref] if mode == "product" else [ref]) + [p for p in sys.path if p and os.path.realpath(p) not in (os.path.realpath(repo), os.path.realpath(ref))]
import torch, yaml
if mode == "reference":
    # leaf_pytorch/filters.py:4 imports torchaudio (absent here) and the default initializer calls melscale_fbanks: a stand-in
    # from this repo's HTK filterbank restatement (initial VALUES are not what this test pins -- keys, shapes, strict loading are)
    import importlib.util
    spec = importlib.util.spec_from_file_location("_oracle_for_caller_test", os.path.join(repo, "oracle", "leaf_oracle.py"))
    lo = importlib.util.module_from_spec(spec); sys.modules[spec.name] = lo; spec.loader.exec_module(lo)
    ta = types.ModuleType("torchaudio"); ta.functional = types.ModuleType("torchaudio.functional")
    def melscale_fbanks(n_freqs, f_min, f_max, n_mels, sample_rate, norm=None, mel_scale="htk"):
        import math
        all_freqs = torch.linspace(0, sample_rate // 2, n_freqs)
        m = lambda f: 2595.0 * math.log10(1.0 + f / 700.0)
        pts = torch.linspace(m(f_min), m(f_max), n_mels + 2)
        f_pts = 700.0 * (10.0 ** (pts / 2595.0) - 1.0)
        diff = f_pts[1:] - f_pts[:-1]
        slopes = f_pts.unsqueeze(0) - all_freqs.unsqueeze(1)
        down, up = -slopes[:, :-2] / diff[:-1], slopes[:, 2:] / diff[1:]
        return torch.clamp(torch.min(down, up), min=0.0)
    ta.functional.melscale_fbanks = melscale_fbanks
    sys.modules["torchaudio"] = ta; sys.modules["torchaudio.functional"] = ta.functional
from models.classifier import Classifier            # the reference's own caller (models/classifier.py:7-18)
import leaf_pytorch
res = {"leaf_pytorch_file": os.path.realpath(leaf_pytorch.__file__), "cfgs": {}}
for path in cfgs:
    cfg = yaml.safe_load(open(path))
    with contextlib.redirect_stdout(io.StringIO()):
        torch.manual_seed(0)
        net = Classifier(cfg)
    fe = net.features
    name = os.path.relpath(path, os.path.join(ref, "cfgs"))
    sd = net.state_dict()
    rec = {"features_class": type(fe).__module__ + "." + type(fe).__qualname__,
           "feature_keys": [k for k in sd if k.startswith("features.")],
           "feature_shapes": {k: list(v.shape) for k, v in sd.items() if k.startswith("features.")},
           "n_filters": int(fe._complex_conv._filters), "kernel_size": int(fe._complex_conv._kernel_size),
           "use_legacy_complex": bool(fe._complex_conv.use_legacy_complex), "n_model_keys": sum(1 for k in sd if k.startswith("model.")),
           "frontend_cfg": cfg["frontend"]}
    sd_path = os.path.join(sd_dir, name.replace(os.sep, "__") + "." + mode + ".pt")
    other = os.path.join(sd_dir, name.replace(os.sep, "__") + "." + ("reference" if mode == "product" else "product") + ".pt")
    if os.path.exists(other):
        # train.py:33-49's checkpoint: {"model_state_dict": ...}; strict load of what the OTHER implementation produced
        ck = torch.load(other)
        missing_unexpected = net.load_state_dict(ck["model_state_dict"], strict=True)
        rec["strict_load_of_the_other"] = str(missing_unexpected)
        back = net.state_dict()
        rec["round_trip_equal"] = all(torch.equal(back[k], ck["model_state_dict"][k]) for k in ck["model_state_dict"])
    torch.save({"model_state_dict": sd}, sd_path)
    res["cfgs"][name] = rec
json.dump(res, open(out_path, "w"))
'''


def run_child(mode, tmp, cfgs):
    out = os.path.join(tmp, f"{mode}.json")
    env = {k: v for k, v in os.environ.items() if k != "PYTHONPATH"}
    r = subprocess.run([sys.executable, "-B", "-c", CHILD, mode, REPO, REFERENCE, out, tmp] + list(cfgs),
                       capture_output=True, text=True, env=env, timeout=900, cwd=tmp)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    return json.load(open(out))


def test_reference_classifier_builds_on_the_product_for_every_shipped_cfg():
    assert len(CFGS) >= 14, CFGS
    with tempfile.TemporaryDirectory() as tmp:
        ref = run_child("reference", tmp, CFGS)          # the reference's Classifier on the reference's Leaf: checkpoints to load
        prod = run_child("product", tmp, CFGS)           # the reference's Classifier on the product (shim ahead on sys.path)
        back = run_child("reference", tmp, CFGS)         # ... and the product's checkpoints back into the reference
    assert prod["leaf_pytorch_file"].startswith(os.path.realpath(REPO) + os.sep), prod["leaf_pytorch_file"]
    assert ref["leaf_pytorch_file"].startswith(os.path.realpath(REFERENCE) + os.sep), ref["leaf_pytorch_file"]
    assert set(prod["cfgs"]) == set(ref["cfgs"]) and len(prod["cfgs"]) == len(CFGS)
    saw_64 = False
    for name, p in prod["cfgs"].items():
        r = ref["cfgs"][name]
        assert p["features_class"] == "leaf_pytorch_amd.frontend.Leaf", (name, p["features_class"])
        assert r["features_class"] == "leaf_pytorch.frontend.Leaf", (name, r["features_class"])
        assert p["feature_keys"] == r["feature_keys"] == SEVEN, (name, p["feature_keys"])
        assert p["feature_shapes"] == r["feature_shapes"], name
        fc = p["frontend_cfg"]
        want_f = 40 if fc.get("default_args", False) else int(fc.get("n_filters", 40))
        assert p["n_filters"] == r["n_filters"] == want_f and p["kernel_size"] == r["kernel_size"] == 401, name
        assert p["use_legacy_complex"] == r["use_legacy_complex"] == bool(fc.get("use_legacy_complex", False)), name
        assert p["n_model_keys"] == r["n_model_keys"] > 0, name
        saw_64 = saw_64 or want_f == 64
        assert p["strict_load_of_the_other"] == "<All keys matched successfully>" and p["round_trip_equal"], (name, p)
        b = back["cfgs"][name]
        assert b["strict_load_of_the_other"] == "<All keys matched successfully>" and b["round_trip_equal"], (name, b)
    assert saw_64, "no shipped cfg with n_filters: 64 was exercised (the AudioSet cfgs)"
