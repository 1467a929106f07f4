"""``Leaf`` -- the drop-in for ``leaf_pytorch.frontend.Leaf`` (reference leaf_pytorch/frontend.py:22-89).

Same constructor signature (order and defaults), same sub-module attribute names, same ``state_dict``
keys and shapes, same ``(B,1,T) -> (B,F,T')`` forward, same error conventions -- but ``forward`` is one
call into the fused HIP kernels for MI355X (include/leaf_hip.h: ``leaf_forward_f32``).  There is no
CPU / eager fallback: a non-HIP input raises.
"""
from __future__ import annotations

import torch
from torch import nn

from . import _native, _ops
from .initializers import GaborInit
from .modules import GaborConv1d, GaussianLowPass, PCENLayer, _SquaredModulusFn


class SquaredModulus(nn.Module):
    """frontend.py:10-19 -- (B,2F,T) interleaved re/im -> (B,F,T) re^2+im^2 (stage kernel when used alone;
    differentiable like the reference's)."""

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        if torch.is_grad_enabled() and x.requires_grad:
            return _SquaredModulusFn.apply(x)
        return _native.squared_modulus(x)


class _LeafForward(torch.autograd.Function):
    """Forward = fused HIP path (leaf_forward_f32); backward = leaf_backward_f32 (recomputes on device)."""

    @staticmethod
    def forward(ctx, x, kernel, pool_w, pool_b, alpha, delta, root, ema_w, K, hop, pcen, algo):
        if algo & _native.OPT_PEAKNORM:
            # leaf_forward_save_f32 would hand the backward a pooled tensor of the NORMALISED clips next to the raw x
            raise RuntimeError("the folded PeakNormalization prologue is forward-only (no backward through it)")
        out, raw = _native.leaf_forward(x, kernel, pool_w, pool_b, alpha, delta, root, ema_w, K, hop, pcen=pcen,
                                        algo=algo, save_raw=True)
        ctx.save_for_backward(x, kernel, pool_w, pool_b, raw, *([alpha, delta, root, ema_w] if pcen else []))
        ctx.geom = (K, hop, pcen)
        ctx.full = bool(algo & _native.ALGO_FULL_TRANSFORMS)      # Leaf.full_transforms(): the backward keeps them too
        ctx.strict = bool(algo & _native.ALGO_STRICT_BAND_CLASSES)
        return out

    @staticmethod
    @torch.autograd.function.once_differentiable      # (gradients of gradients: the dispatcher-op path, _second_order.py; here they raise)
    def backward(ctx, grad_out):
        K, hop, pcen = ctx.geom
        saved = ctx.saved_tensors
        x, kernel, pool_w, pool_b, raw = saved[:5]
        alpha, delta, root, ema_w = saved[5:] if pcen else (None,) * 4
        need_dx = ctx.needs_input_grad[0]
        gk, gpw, gpb, ga, gd, gr, gw, gx = _native.leaf_backward(x, kernel, pool_w, pool_b, alpha, delta, root, ema_w, K, hop,
                                                                 grad_out, pcen=pcen, need_dx=need_dx, pooled_raw=raw,
                                                                 full_transforms=ctx.full, strict_band_classes=ctx.strict)
        if gx is not None:
            gx = gx.reshape(x.shape)
        return gx, gk, gpw, gpb, ga, gd, gr, gw, None, None, None, None


class Leaf(nn.Module):
    def __init__(self, n_filters: int = 40, sample_rate: int = 16000, window_len: float = 25.,
                 window_stride: float = 10., preemp: bool = False, init_min_freq=60.0, init_max_freq=7800.0,
                 mean_var_norm: bool = False, pcen_compression: bool = True, use_legacy_complex=False,
                 initializer="default"):
        super().__init__()
        window_size = int(sample_rate * window_len // 1000 + 1)
        window_stride = int(sample_rate * window_stride // 1000)
        if preemp:
            raise NotImplementedError("Pre-emp functionality not implemented yet..")
        self._preemp = None
        if initializer == "default":
            initializer = GaborInit(default_window_len=window_size, sample_rate=sample_rate,
                                    min_freq=init_min_freq, max_freq=init_max_freq)
        self._complex_conv = GaborConv1d(filters=2 * n_filters, kernel_size=window_size, strides=1, padding="same",
                                         use_bias=False, initializer=initializer,
                                         use_legacy_complex=use_legacy_complex)
        self._activation = SquaredModulus()
        self._pooling = GaussianLowPass(n_filters, kernel_size=window_size, strides=window_stride, padding="same")
        self._instance_norm = None
        if mean_var_norm:
            raise NotImplementedError("Instance Norm functionality not added yet..")
        if pcen_compression:
            self._compression = PCENLayer(n_filters, alpha=0.96, smooth_coef=0.04, delta=2.0, floor=1e-12,
                                          trainable=True, learn_smooth_coef=True, per_channel_smooth_coef=True)
        else:
            self._compression = None
        self._maximum_val = torch.tensor(1e-5)
        self._algo = _native.ALGO_AUTO       # not part of the reference surface: kernel selector for tests/bench
        self._cache_tables = False           # not part of the reference surface: see cache_tables()
        self._tables = None
        self._tables_key = None
        self._fuse_peaknorm = False          # not part of the reference surface: see fuse_peak_normalization()

    def full_transforms(self, enable: bool = True) -> "Leaf":
        """Not part of the reference surface: switch the band-limited filter tasks off for this module -- every filter on the
        full-length inverse transform in the forward (LEAF_ALGO_FULL_TRANSFORMS) AND in the backward the autograd path runs
        (LEAF_FLAG_BWD_FULL_TRANSFORMS).  The default (band tasks where a filter's spectrum allows) stays within ~1e-6 of the
        oracle in the forward and ~1e-5 of the full-transform gradients; this is the opt-out for a caller who wants the
        formulation without the approximation, at ~1.8x the time of the 16 kHz forward."""
        self._algo = (self._algo | _native.ALGO_FULL_TRANSFORMS) if enable else (self._algo & ~_native.ALGO_FULL_TRANSFORMS)
        return self

    def fuse_peak_normalization(self, enable: bool = True) -> "Leaf":
        """Not part of the reference surface: make ``forward(x)`` return ``Leaf(PeakNormalization(x))`` -- the last transform
        of every reference data pipeline (utilities/data/raw_transforms.py:334-345) folded into the frontend.  On the
        overlap-save paths under ``no_grad`` the normalised waveform is never written (one read-only pass finds each clip's
        scale s; s^2 multiplies the pooled energies where the bias is added: LEAF_FLAG_PEAKNORM); under autograd, or where
        another kernel family serves the geometry, the separate HIP normalisation kernel runs first.  Equal to the two-step
        form up to fp32 rounding."""
        self._fuse_peaknorm = bool(enable)
        return self

    def cache_tables(self, enable: bool = True) -> "Leaf":
        """Serving mode (not part of the reference surface): keep the tables derived from the filter / pooling parameters
        (filter spectra, pooling rows) across no-grad forwards instead of rebuilding them on every call, and rebuild them
        only when those parameters change (torch's per-tensor version counter and storage pointer are checked on every
        call, so optimizer steps, ``load_state_dict`` and ``.to()`` are picked up; writes that bypass the version counter
        are not).  Outputs are bit-identical to the default path."""
        self._cache_tables = bool(enable)
        self._tables = self._tables_key = None
        return self

    def _prepared_tables(self):
        k, w = self._complex_conv._kernel, self._pooling.weights
        key = (k.data_ptr(), k._version, w.data_ptr(), w._version, k.device)
        if self._tables is None or key != self._tables_key:
            self._tables = _native.prepare_tables(k.detach(), w.detach(), self._complex_conv._kernel_size, self._pooling.strides)
            self._tables_key = key if self._tables is not None else None
        return self._tables

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        _native.require_hip(x, "Leaf.forward")
        algo = self._algo
        c = self._compression
        params = (self._complex_conv._kernel, self._pooling.weights, self._pooling._bias,
                  c.alpha if c is not None else None, c.delta if c is not None else None,
                  c.root if c is not None else None, c.ema._weights if c is not None else None)
        # from the tensors actually handed to the kernel, not self.parameters(): nn.DataParallel replicas hold plain
        # (non-leaf) tensors, for which parameters() is empty
        params_need_grad = torch.is_grad_enabled() and any(t is not None and t.requires_grad for t in params)
        if self._fuse_peaknorm:
            K_, hop_ = self._complex_conv._kernel_size, self._pooling.strides
            # the folded prologue is forward-only: the same condition that sends the call to the training path below
            fused_ok = (not (params_need_grad or (torch.is_grad_enabled() and x.requires_grad))
                        and not self._cache_tables and x.dim() == 3 and x.shape[1] == 1 and x.shape[0] > 0)
            if fused_ok:
                with torch.cuda.device(x.device):
                    sel = algo & 0xff
                    if sel == _native.ALGO_AUTO:
                        sel = _native.load().leaf_auto_algo(x.shape[0], x.shape[-1], self._complex_conv._filters, K_, hop_)
                fused_ok = sel in (_native.ALGO_FFT, _native.ALGO_FFT_WG, _native.ALGO_FFT_SMALL)
            if fused_ok:
                algo = algo | _native.OPT_PEAKNORM
            else:
                from .transforms import PeakNormalization
                x = PeakNormalization()(x)
        if c is not None and c._floor != 1e-12:
            raise NotImplementedError("fused path is specialised for the PCEN floor Leaf constructs (1e-12)")
        args = (x, *params, self._complex_conv._kernel_size, self._pooling.strides, c is not None, algo)
        needs_grad = params_need_grad or (torch.is_grad_enabled() and x.requires_grad)
        if needs_grad and (algo & _native.OPT_PEAKNORM):
            raise RuntimeError("Leaf.forward: the folded PeakNormalization prologue is forward-only")   # unreachable by design
        if _ops.available():
            # dispatcher ops (csrc/torch_binding.cpp): traceable by torch.compile / export, one hop per eager call
            _ops.load()
            if needs_grad:
                if x.dtype == torch.bfloat16:
                    raise RuntimeError("Leaf.forward: the backward is float32 only -- bfloat16 I/O is an inference extension")
                return _ops.forward_train(*args[:10], algo=args[11])
            if not (self._cache_tables and not torch.compiler.is_compiling()):
                return _ops.forward(*args[:10], algo=args[11])
        if needs_grad:
            return _LeafForward.apply(*args)
        if self._cache_tables and self._algo in (_native.ALGO_AUTO, _native.ALGO_FFT):
            K, hop = args[8], args[9]
            B, T, F = x.shape[0], x.shape[-1], args[1].shape[0]
            with torch.cuda.device(x.device):            # the plan is sized for the CU count of the device the call runs on
                plan = _native.fft_plan_info(B, T, F, K, hop)
                auto = _native.load().leaf_auto_algo(B, T, F, K, hop)
            # the prepared tables are those of the 2048-sample plan: used only where that is what the default path runs, so
            # that serving mode stays bit-identical to it (long windows on 4096-sample blocks rebuild their tables per call)
            if auto in (_native.ALGO_FFT, _native.ALGO_FFT_WG) and plan is not None and plan["fft_n"] == 2048:
                tables = self._prepared_tables()
                if tables is not None:
                    return _native.leaf_forward_prepared(x, tables, args[3], args[4], args[5], args[6], args[7],
                                                         args[1].shape[0], K, hop, pcen=args[10])
        return _native.leaf_forward(*args[:8], args[8], args[9], pcen=args[10], algo=args[11] & ~_native.OPT_PEAKNORM,
                                    peak_normalize=bool(args[11] & _native.OPT_PEAKNORM))
