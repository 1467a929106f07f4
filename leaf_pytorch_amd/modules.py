"""The four operator modules of the LEAF frontend, with the reference's names, constructor arguments,
parameter names/shapes/initial values and error behaviour -- and HIP kernels (through the C ABI in
include/leaf_hip.h) as their forward arithmetic.

Reference counterparts:
    GaborConstraint, GaborConv1d   leaf_pytorch/convolution.py:10-22, 25-99
    GaussianLowPass                leaf_pytorch/pooling.py:8-42
    ExponentialMovingAverage       leaf_pytorch/postprocessing.py:5-28
    PCENLayer                      leaf_pytorch/postprocessing.py:31-69

Inside ``Leaf.forward`` these forwards are NOT called: the fused kernel reads the parameters these
modules own.  Calling a sub-module on its own runs the corresponding stage kernel; like the reference's modules they
are differentiable (first order): each stage is a ``torch.autograd.Function`` whose backward is the matching
``leaf_*_backward_f32`` entry point of the C ABI, so a training script that composes the sub-modules itself gets the
same gradients the reference's stock-op graph yields (tests/test_gpu_backward.py).

Two limits of the stand-alone stages, both deliberate:
  * first order only -- the stage Functions are ``once_differentiable``: asking for a gradient of a gradient of a stand-alone
    stage raises instead of returning something silently wrong (``Leaf`` itself supports ``create_graph=True`` since round 6:
    _second_order.py);
  * their backward kernels are the plain per-stage ones of the staged path (one thread per tap over all T samples for the
    tap gradients, a serial loop over B*T' per (filter, tap) for the pooling window): correct, checked against fp64
    autograd, and orders of magnitude slower than the fused backward at training batch sizes.  A training script should
    call ``Leaf`` (0.7 ms per step at 256 x 1 s); compose the sub-modules only for inspection or small inputs.
"""
from __future__ import annotations

import math
from typing import Callable

import torch
from torch import nn

from . import _native


class _GaborConvFn(torch.autograd.Function):
    """convolution.py:71-99 as one differentiable op: forward leaf_gabor_conv_f32, backward leaf_gabor_conv_backward_f32."""

    @staticmethod
    def forward(ctx, x, kernel, K):
        ctx.save_for_backward(x, kernel)
        ctx.K = K
        return _native.gabor_conv(x, kernel, K)

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, grad_y):
        x, kernel = ctx.saved_tensors
        gk, gx = _native.gabor_conv_backward(x, kernel, ctx.K, grad_y.contiguous(), need_dk=ctx.needs_input_grad[1],
                                             need_dx=ctx.needs_input_grad[0])
        return gx, gk, None


class _SquaredModulusFn(torch.autograd.Function):
    """frontend.py:15-19."""

    @staticmethod
    def forward(ctx, y):
        ctx.save_for_backward(y)
        return _native.squared_modulus(y)

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, grad_e):
        (y,) = ctx.saved_tensors
        return _native.squared_modulus_backward(y, grad_e.contiguous())


class _GaussianLowPassFn(torch.autograd.Function):
    """pooling.py:31-42 (window from impulse_responses.py:74-80)."""

    @staticmethod
    def forward(ctx, e, pool_w, pool_b, K, hop):
        ctx.save_for_backward(e, pool_w)
        ctx.geom = (K, hop, pool_b is not None)
        return _native.gaussian_lowpass(e, pool_w, pool_b, K, hop)

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, grad_pooled):
        e, pool_w = ctx.saved_tensors
        K, hop, has_bias = ctx.geom
        ge, gw, gb = _native.gaussian_lowpass_backward(e, pool_w, K, hop, grad_pooled.contiguous(),
                                                       need_de=ctx.needs_input_grad[0], need_dw=ctx.needs_input_grad[1],
                                                       need_db=has_bias and ctx.needs_input_grad[2])
        return ge, gw, gb, None, None


class _EmaFn(torch.autograd.Function):
    """postprocessing.py:13-28."""

    @staticmethod
    def forward(ctx, p, ema_w):
        ctx.save_for_backward(p, ema_w)
        return _native.ema(p, ema_w)

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, grad_ema):
        p, ema_w = ctx.saved_tensors
        return _native.ema_backward(p, ema_w, grad_ema.contiguous())


class _PcenFn(torch.autograd.Function):
    """postprocessing.py:62-69."""

    @staticmethod
    def forward(ctx, p, alpha, delta, root, ema_w, floor):
        ctx.save_for_backward(p, alpha, delta, root, ema_w)
        ctx.floor = floor
        return _native.pcen(p, alpha, delta, root, ema_w, floor)

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, grad_out):
        p, alpha, delta, root, ema_w = ctx.saved_tensors
        return (*_native.pcen_backward(p, alpha, delta, root, ema_w, ctx.floor, grad_out.contiguous()), None)


def _wants_grad(*tensors) -> bool:
    return torch.is_grad_enabled() and any(t is not None and t.requires_grad for t in tensors)


def get_padding_value(kernel_size: int):
    """utils.py:5-10 -- ("same") padding pair (left, right)."""
    return kernel_size // 2 + kernel_size % 2 - 1, kernel_size // 2


class GaborConstraint(nn.Module):
    """Clamp mu to [0, pi] and sigma to [4c, Kc], c = sqrt(2 ln 2)/pi (functional; parameter untouched).

    The kernels apply the same clamp internally; this module exists for API parity and host-side use.
    """

    def __init__(self, kernel_size: int):
        super().__init__()
        self._kernel_size = kernel_size

    def forward(self, kernel_data: torch.Tensor) -> torch.Tensor:
        # convolution.py:18-19 builds both sigma bounds from a float32 TENSOR (sqrt(2 log 2) rounded to fp32, then / pi and
        # x 4 or x K in fp32) -- not from Python doubles: 4c and K c differ from the float64 values in the last bit, and
        # the kernels (leaf_common.hpp gabor_bounds) and the oracle (constrain_gabor) use the fp32-built ones.
        c32 = torch.sqrt(2.0 * torch.log(torch.tensor(2.0, device=kernel_data.device))) / math.pi
        mu = torch.clamp(kernel_data[:, 0], 0.0, math.pi)
        sigma = torch.clamp(kernel_data[:, 1], 4 * c32, self._kernel_size * c32)
        return torch.stack([mu, sigma], dim=-1)


class GaborConv1d(nn.Module):
    def __init__(self, filters, kernel_size, strides, padding, initializer=None, use_bias=False,
                 sort_filters=False, use_legacy_complex=False):
        super().__init__()
        self._filters = filters // 2
        self._kernel_size = kernel_size
        self._strides = strides
        self._padding = padding
        self._use_bias = use_bias
        self._sort_filters = sort_filters
        shape = (self._filters, 2)
        if isinstance(initializer, Callable):
            init_weights = initializer(shape)
        elif initializer == "random":
            init_weights = torch.randn(*shape)
        elif initializer == "xavier_normal":
            init_weights = nn.init.xavier_normal_(torch.randn(*shape))
        elif initializer == "kaiming_normal":
            init_weights = nn.init.kaiming_normal_(torch.randn(*shape))
        else:
            raise ValueError("unsupported initializer")
        self.constraint = GaborConstraint(self._kernel_size)
        self._kernel = nn.Parameter(init_weights)
        self._pad_value = get_padding_value(self._kernel_size) if self._padding.lower() == "same" else self._padding
        self._bias = nn.Parameter(torch.ones(self._filters * 2)) if self._use_bias else None
        # both tap-synthesis variants of the reference are the same function (they differ <= 3.7e-9);
        # the flag is kept so shipped cfgs (use_legacy_complex: True) load unchanged.
        self.use_legacy_complex = use_legacy_complex

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        if self._sort_filters:
            raise NotImplementedError("sort filter functionality not yet implemented")
        if self._strides != 1 or self._padding.lower() != "same":
            raise NotImplementedError("the HIP GaborConv1d supports strides=1, padding='same' (what Leaf uses)")
        if _wants_grad(x, self._kernel):
            y = _GaborConvFn.apply(x, self._kernel, self._kernel_size)
        else:
            y = _native.gabor_conv(x, self._kernel, self._kernel_size)
        if self._bias is not None:
            y = y + self._bias.view(1, -1, 1)
        return y

    def filters(self) -> torch.Tensor:
        """(2F,K) taps as the convolution consumes them (row 2f = Re, 2f+1 = Im)."""
        return _native.gabor_taps(self._kernel, self._kernel_size)


class GaussianLowPass(nn.Module):
    def __init__(self, in_channels, kernel_size, strides=1, padding="same", use_bias=True):
        super().__init__()
        self.kernel_size = kernel_size
        self.strides = strides
        self.padding = padding
        self.use_bias = use_bias
        self.in_channels = in_channels
        self.weights = nn.Parameter(torch.full((1, 1, in_channels, 1), 0.4))   # 0.4 ~ Hanning window
        self._bias = nn.Parameter(torch.ones(in_channels)) if use_bias else None
        self.pad_value = get_padding_value(kernel_size) if padding.lower() == "same" else padding

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        if self.padding.lower() != "same":
            raise NotImplementedError("the HIP GaussianLowPass supports padding='same' (what Leaf uses)")
        if _wants_grad(x, self.weights, self._bias):
            return _GaussianLowPassFn.apply(x, self.weights, self._bias, self.kernel_size, self.strides)
        return _native.gaussian_lowpass(x, self.weights, self._bias, self.kernel_size, self.strides)


class ExponentialMovingAverage(nn.Module):
    def __init__(self, in_channels, coeff_init, per_channel: bool = False):
        super().__init__()
        self._coeff_init = coeff_init
        self._per_channel = per_channel
        self._weights = nn.Parameter(torch.ones(in_channels if per_channel else 1) * coeff_init)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        if _wants_grad(x, self._weights):
            return _EmaFn.apply(x, self._weights)
        return _native.ema(x, self._weights)


class PCENLayer(nn.Module):
    def __init__(self, in_channels, alpha: float = 0.96, smooth_coef: float = 0.04, delta: float = 2.0,
                 root: float = 2.0, floor: float = 1e-6, trainable: bool = False,
                 learn_smooth_coef: bool = False, per_channel_smooth_coef: bool = False):
        super().__init__()
        self._alpha_init, self._delta_init, self._root_init = alpha, delta, root
        self._smooth_coef = smooth_coef
        self._floor = floor
        self._trainable = trainable
        self._learn_smooth_coef = learn_smooth_coef
        self._per_channel_smooth_coef = per_channel_smooth_coef
        self.alpha = nn.Parameter(torch.ones(in_channels) * alpha)
        self.delta = nn.Parameter(torch.ones(in_channels) * delta)
        self.root = nn.Parameter(torch.ones(in_channels) * root)
        if not learn_smooth_coef:
            raise ValueError("SimpleRNN based ema not implemented.")
        self.ema = ExponentialMovingAverage(in_channels, coeff_init=smooth_coef, per_channel=per_channel_smooth_coef)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        if _wants_grad(x, self.alpha, self.delta, self.root, self.ema._weights):
            return _PcenFn.apply(x, self.alpha, self.delta, self.root, self.ema._weights, self._floor)
        return _native.pcen(x, self.alpha, self.delta, self.root, self.ema._weights, self._floor)
