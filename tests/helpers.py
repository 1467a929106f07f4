"""Shared helpers for the parity tests (kept free of any product logic)."""
import torch

from leaf_pytorch_amd import Leaf


def make_leaf(n_filters, window_size, hop, pcen, params=None, device=None):
    """Build the product Leaf with an exact (K, hop) geometry: sample_rate=1000 makes
    K = window_len + 1 and hop = window_stride (frontend.py:38-39 integer arithmetic)."""
    kernel = params["_complex_conv._kernel"] if params is not None else torch.rand(n_filters, 2)
    m = Leaf(n_filters=n_filters, sample_rate=1000, window_len=float(window_size - 1), window_stride=float(hop),
             pcen_compression=pcen, initializer=lambda shape: kernel.clone())
    assert m._complex_conv._kernel_size == window_size and m._pooling.strides == hop
    if params is not None:
        m.load_state_dict(params, strict=True)
    m.eval()
    for p in m.parameters():
        p.requires_grad_(False)
    return m.to(device) if device is not None else m
