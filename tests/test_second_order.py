"""Gradients of gradients (leaf_pytorch_amd/_second_order.py; VERDICT r5 "missing" #4): the reference's stock-op graph
(leaf_pytorch/frontend.py:78-89) supports ``create_graph=True``; here ``leaf_amd::backward`` gets an autograd formula of its own.

CPU (no GPU needed): the differentiable composite restatement of the forward against the oracle (forward values and first-order
gradients in fp64), and the formula itself -- cotangents on the first-order gradients in, second-order gradients out -- against
autograd-of-autograd through the oracle.  GPU: the same quantity through ``Leaf`` (HIP forward, HIP first-order backward, formula
on the device) against fp64 autograd-of-autograd through the oracle.
"""
import types

import pytest
import torch

from oracle import leaf_oracle as lo

NAMES = ["_complex_conv._kernel", "_pooling.weights", "_pooling._bias", "_compression.alpha", "_compression.delta",
         "_compression.root", "_compression.ema._weights"]


def _case(F, K, hop, T, B, pcen, seed):
    gen = torch.Generator().manual_seed(seed)
    geo = lo.LeafGeometry(F, 0, K, hop, *lo.same_padding(K))
    params = lo.default_params(geo, pcen, kernel=torch.stack([0.2 + torch.rand(F, generator=gen) * 2.5,
                                                              5.0 + torch.rand(F, generator=gen) * K / 5], 1))
    params = {k: v * (1 + 0.1 * (2 * torch.rand(v.shape, generator=gen) - 1)) for k, v in params.items() if pcen or "_compression" not in k}
    x = torch.randn(B, 1, T, generator=gen)
    return geo, params, x


def _oracle_second_order(x, params, geo, pcen, with_x):
    """P = sum |dL/dtheta|^2 (+ |dL/dx|^2), L = <out, go>: (dP/dtheta, dP/dx), the first-order gradients and go, all in fp64."""
    leaves = {k: v.double().clone().requires_grad_(True) for k, v in params.items()}
    xx = x.double().clone().requires_grad_(True)
    out = lo.leaf_forward(xx, leaves, geo, pcen, torch.float64)
    go = torch.cos(torch.arange(out.numel(), dtype=torch.float64)).reshape(out.shape)
    wrt = [*leaves.values()] + ([xx] if with_x else [])
    gs = torch.autograd.grad(out, wrt, go, create_graph=True)
    P = sum((g * g).sum() for g in gs)
    second = torch.autograd.grad(P, [*leaves.values(), xx], allow_unused=True)    # (None: P does not depend on it, e.g. the bias without PCEN)
    return dict(zip([*leaves.keys(), "x"], second)), [g.detach() for g in gs], go


@pytest.mark.parametrize("pcen", [True, False])
def test_composite_and_formula_match_autograd_of_autograd_through_the_oracle(pcen):
    from leaf_pytorch_amd import _second_order as so
    F, K, hop, T, B = 5, 101, 40, 700, 3
    geo, params, x = _case(F, K, hop, T, B, pcen, 3)
    p64 = {k: v.double() for k, v in params.items()}
    args = [p64[n] for n in NAMES[:3]] + ([p64[n] for n in NAMES[3:]] if pcen else [None] * 4)
    ref = lo.leaf_forward(x.double(), p64, geo, pcen, torch.float64)
    got = so.composite_forward(x.double(), *args, K, hop)
    assert float((got - ref).abs().max() / ref.abs().max()) < 1e-12
    # even windows and a hop that does not divide the clip: the same padding / frame count as the oracle
    geo2, params2, x2 = _case(3, 64, 17, 333, 2, pcen, 4)
    a2 = [params2[n].double() for n in NAMES[:3]] + ([params2[n].double() for n in NAMES[3:]] if pcen else [None] * 4)
    ref2 = lo.leaf_forward(x2.double(), {k: v.double() for k, v in params2.items()}, geo2, pcen, torch.float64)
    got2 = so.composite_forward(x2.double(), *a2, 64, 17)
    assert got2.shape == ref2.shape and float((got2 - ref2).abs().max() / ref2.abs().max()) < 1e-12
    for with_x in (True, False):
        second, G, go = _oracle_second_order(x, params, geo, pcen, with_x)
        ctx = types.SimpleNamespace(pcen=pcen, geom=(K, hop), need_dx=with_x)
        ctx.saved_tensors = (x.double(), args[0], args[1], args[2], go, *(args[3:] if pcen else []))
        npar = 7 if pcen else 3
        grads = [2 * g for g in G[:npar]] + [None] * (7 - npar) + [2 * G[npar] if with_x else None]
        res = so.backward(ctx, grads)
        assert len(res) == 14 and all(res[i] is None for i in (8, 9, 11, 12, 13))
        mine = dict(zip(["x", *NAMES[:npar]], [res[0], *res[1:1 + npar]]))
        for k, r in second.items():
            g = mine[k]
            if r is None:
                assert g is None or float(g.abs().max()) == 0.0, k
                continue
            assert float((g - r.reshape(g.shape)).abs().max() / r.abs().max()) < 1e-10, (k, with_x)
        # d<v, G>/d grad_out = J v: checked through <J v, w> = <v, J^T w> with w = go (J^T go = G, v = 2 G)
        assert abs(float((res[10] * go).sum()) - 2 * float(sum((g * g).sum() for g in G))) < 1e-8 * float(sum((g * g).sum() for g in G))


@pytest.mark.gpu
@pytest.mark.parametrize("pcen,with_x", [(True, False), (True, True), (False, True)])
def test_gradient_penalty_through_leaf_matches_the_oracle(pcen, with_x):
    """HIP forward + HIP first-order backward + the formula on the device: d/dtheta of sum |dL/dtheta|^2 (+ |dL/dx|^2)."""
    from test_gpu_backward import make_leaf
    F, K, hop, T, B = 6, 401, 160, 3300, 4
    geo, params, x = _case(F, K, hop, T, B, pcen, 11)
    second, G, go = _oracle_second_order(x, params, geo, pcen, with_x)
    m = make_leaf(F, K, hop, pcen, params, "cuda:0")
    for p in m.parameters():
        p.requires_grad_(True)
    xd = x.to("cuda:0").requires_grad_(with_x)
    out = m(xd)
    named = dict(m.named_parameters())
    wrt = [named[k] for k in params] + ([xd] if with_x else [])
    gs = torch.autograd.grad(out, wrt, go.float().to("cuda:0"), create_graph=True)
    for g, r in zip(gs, G):                                       # the first order is still the HIP backward's
        assert float((g.detach().cpu().double().reshape(r.shape) - r).abs().max() / r.abs().max()) < 1e-4
    P = sum((g * g).sum() for g in gs)
    got = torch.autograd.grad(P, [named[k] for k in params] + ([xd] if with_x else []), allow_unused=True)
    keys = [*params.keys()] + (["x"] if with_x else [])
    for k, g in zip(keys, got):
        r = second[k]
        if r is None:
            assert g is None or float(g.abs().max()) == 0.0, k
            continue
        assert g is not None, k
        err = float((g.cpu().double().reshape(r.shape) - r).abs().max() / r.abs().max())
        assert err < 2e-3, (k, err)                                # fp32 first-order gradients as cotangents, fp32 composite
