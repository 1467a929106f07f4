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


# ---- gradient comparison ----------------------------------------------------------------------------
# A gradient tensor is compared COLUMN by column (mu and sigma of `_complex_conv._kernel` (F, 2) separately; the (F,) tensors
# and `_pooling.weights` (1, 1, F, 1) are one column), because the mu column of the kernel is ~650x its sigma column at the
# default parameters: against the tensor's largest entry a wrong d/d sigma passes.  Two bounds, both must hold:
#   (A)  |g - r| <= GRAD_COL_TOL * max|r_col|                        every entry of the column
#   (B)  |g - r| <= GRAD_ENTRY_RTOL * |r_f| + GRAD_ENTRY_ATOL * max|r_col|   per filter f
# r = fp64 autograd through the oracle.  (B) holds each filter's own gradient to three digits unless it is below a millionth of
# the column's largest (fp32 sums over B * T samples cannot resolve less).
import json
import os

GRAD_COL_TOL = float(os.environ.get("LEAF_TEST_GRAD_TOL", "1e-4"))
GRAD_ENTRY_RTOL = float(os.environ.get("LEAF_TEST_GRAD_ENTRY_RTOL", "1e-3"))
GRAD_ENTRY_ATOL = float(os.environ.get("LEAF_TEST_GRAD_ENTRY_ATOL", "1e-6"))
_GRAD_LOG = os.environ.get("LEAF_GRAD_LOG")          # JSON lines of every comparison's worst figures (profiles/r06/)


def grad_columns(name, t):
    """The columns a gradient tensor is judged by: [(label, 1-D view)]."""
    if t.dim() == 2 and t.shape[1] == 2 and "kernel" in name:
        return [(name + "[mu]", t[:, 0]), (name + "[sigma]", t[:, 1])]
    return [(name, t.reshape(-1))]


def grad_errors(name, g, r):
    """Worst figures per column: (label, err_A = max|g-r| / max|r_col|, err_B = max (|g-r| / (rtol |r| + atol max|r_col|)))."""
    g = g.detach().cpu().double().reshape(r.shape)
    r = r.detach().cpu().double()
    rows = []
    for (label, gc), (_, rc) in zip(grad_columns(name, g), grad_columns(name, r)):
        if rc.numel() == 0:
            rows.append((label, 0.0, 0.0))
            continue
        top = float(rc.abs().max())
        d = (gc - rc).abs()
        if top == 0.0:                                   # an all-zero column (e.g. every sigma on a clamp): exact zeros asked
            bad = float(d.max())
            rows.append((label, bad, bad / GRAD_ENTRY_ATOL if bad else 0.0))
            continue
        err_a = float(d.max()) / top
        err_b = float((d / (GRAD_ENTRY_RTOL * rc.abs() + GRAD_ENTRY_ATOL * top)).max())
        rows.append((label, err_a, err_b))
    return rows


def assert_grad_close(name, g, r, ctx="", col_tol=None, entrywise=True):
    """Both bounds above on every column of one gradient tensor; the message names the column and the measured figures.
    `entrywise=False` (dL/dx only: a million entries, most of them tiny sums of forty filters' shares) keeps bound (A)."""
    tol = GRAD_COL_TOL if col_tol is None else col_tol
    rows = grad_errors(name, g, r)
    if _GRAD_LOG:
        with open(_GRAD_LOG, "a") as fh:
            for label, ea, eb in rows:
                fh.write(json.dumps({"column": label, "rel_to_col_max": ea, "entry_bound_used": eb if entrywise else None,
                                     "ctx": str(ctx)}) + "\n")
    for label, ea, eb in rows:
        assert ea < tol, f"{label}: {ea:.3e} of the column's largest entry (bound {tol:.0e}) {ctx}"
        if entrywise:
            assert eb < 1.0, (f"{label}: an entry is off by {eb:.2f}x the per-filter bound "
                              f"({GRAD_ENTRY_RTOL:.0e} |r_f| + {GRAD_ENTRY_ATOL:.0e} max|r_col|); column figure {ea:.3e} {ctx}")
    return rows
