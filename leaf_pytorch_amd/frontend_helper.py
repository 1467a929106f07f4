"""``get_frontend(cfg)`` -- reference leaf_pytorch/frontend_helper.py:7-54: config dict -> ``Leaf``.

Reads the same keys (``frontend.{name,default_args,use_legacy_complex,initializer,n_filters,min_freq,
max_freq,pcen_compress,mean_var_norm,preemp,pretrained}``, ``audio_config.{sample_rate,window_len,
window_stride}``) with the same defaults and the same ``bool(...)`` coercions, so
``models.classifier.Classifier(cfg)`` builds the HIP frontend from an unmodified cfg.
"""
import os

import torch

from .frontend import Leaf


def get_frontend(opt):
    fe_cfg, audio_cfg = opt['frontend'], opt['audio_config']
    pretrained = fe_cfg.get("pretrained", "")
    state = torch.load(pretrained) if os.path.isfile(pretrained) else None
    if "leaf" not in fe_cfg['name'].lower():
        raise NotImplementedError("Other front ends not implemented yet.")
    common = dict(use_legacy_complex=fe_cfg.get("use_legacy_complex", False),
                  initializer=fe_cfg.get("initializer", "default"))
    if fe_cfg.get("default_args", False):
        fe = Leaf(**common)
    else:
        fe = Leaf(n_filters=int(fe_cfg.get("n_filters", 40.0)),
                  sample_rate=int(audio_cfg.get("sample_rate", 16000)),
                  window_len=float(audio_cfg.get("window_len", 25.)),
                  window_stride=float(audio_cfg.get("window_stride", 10.)),
                  preemp=bool(fe_cfg.get("preemp", False)),
                  init_min_freq=float(fe_cfg.get("min_freq", 60.0)),
                  init_max_freq=float(fe_cfg.get("max_freq", 7800.0)),
                  mean_var_norm=bool(fe_cfg.get("mean_var_norm", False)),
                  pcen_compression=bool(fe_cfg.get("pcen_compress", True)),
                  **common)
    if state is not None:
        fe.load_state_dict(state)
    return fe
