"""Dispatcher ops ``torch.ops.leaf_amd.{forward, forward_train, backward}`` (csrc/torch_binding.cpp: a thin torch
extension over the C ABI of include/leaf_hip.h) plus what only Python can attach to them: the fake (meta) kernels
``torch.compile`` / ``torch.export`` need to propagate shapes, and the autograd formula of the training forward.

``Leaf.forward`` goes through these ops, so a model containing the frontend compiles without a graph break and an eager
call costs one dispatcher hop.  There is no fallback: the ops exist only for HIP tensors and fail loudly otherwise.
"""
from __future__ import annotations

import os
import subprocess
import sysconfig
import threading
from typing import Optional

import torch

from . import _native

OPS_LIB_PATH = os.path.join(_native._PKG_DIR, "_leaf_torch_ops.so")
OPS_SRC_PATH = os.path.join(_native._PKG_DIR, "csrc", "torch_binding.cpp")
_loaded = False
_load_lock = threading.Lock()


def build(force: bool = False, verbose: bool = False) -> str:
    """Compile csrc/torch_binding.cpp (plain C++, g++) against the torch headers and libleaf_hip.so, in-tree."""
    from torch.utils import cpp_extension as ce
    deps = [OPS_SRC_PATH, os.path.join(_native.INCLUDE_DIR, "leaf_hip.h")]
    if not force and os.path.exists(OPS_LIB_PATH) and os.path.getmtime(OPS_LIB_PATH) >= max(os.path.getmtime(d) for d in deps):
        return OPS_LIB_PATH
    _native.build()                                          # links against libleaf_hip.so
    tl = os.path.join(os.path.dirname(torch.__file__), "lib")
    inc = ce.include_paths() + ["/opt/rocm/include", _native.INCLUDE_DIR, sysconfig.get_paths()["include"]]
    cmd = [os.environ.get("CXX", "g++"), "-O2", "-std=c++17", "-fPIC", "-shared", "-DUSE_ROCM", "-D__HIP_PLATFORM_AMD__",
           f"-D_GLIBCXX_USE_CXX11_ABI={int(torch._C._GLIBCXX_USE_CXX11_ABI)}"] + [f"-I{i}" for i in inc] + [
           OPS_SRC_PATH, "-o", OPS_LIB_PATH + ".tmp", f"-L{tl}", "-ltorch", "-ltorch_cpu", "-lc10", "-lc10_hip",
           f"-L{_native._PKG_DIR}", "-lleaf_hip", "-Wl,-rpath,$ORIGIN", f"-Wl,-rpath,{tl}"]
    if verbose:
        print(" ".join(cmd))
    subprocess.run(cmd, check=True)
    os.replace(OPS_LIB_PATH + ".tmp", OPS_LIB_PATH)
    return OPS_LIB_PATH


def load() -> None:
    """Register the ops (idempotent).  Raises if the extension has not been built -- no fallback."""
    global _loaded
    if _loaded:
        return
    # The first call may come from several threads at once (nn.DataParallel replicas call Leaf.forward concurrently, and
    # dlopen releases the GIL): register exactly once.
    with _load_lock:
        if _loaded:
            return
        if not os.path.exists(OPS_LIB_PATH):
            raise RuntimeError(f"{OPS_LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'`")
        _native.load()
        torch.ops.load_library(OPS_LIB_PATH)
        _register_python_side()
        _loaded = True


def available() -> bool:
    return _loaded or os.path.exists(OPS_LIB_PATH)


def _frames(T: int, K: int, hop: int) -> int:
    pad_l, pad_r = K // 2 + K % 2 - 1, K // 2                 # utils.py:5-10
    return (T + pad_l + pad_r - K) // hop + 1


def _register_python_side() -> None:
    @torch.library.register_fake("leaf_amd::forward")
    def _(x, kernel, pool_w, pool_b, alpha, delta, root, ema_w, K, hop, log1p, algo):
        return x.new_empty((x.shape[0], kernel.shape[0], _frames(x.shape[-1], K, hop)))

    @torch.library.register_fake("leaf_amd::forward_train")
    def _(x, kernel, pool_w, pool_b, alpha, delta, root, ema_w, K, hop, algo):
        shape = (x.shape[0], kernel.shape[0], _frames(x.shape[-1], K, hop))
        return x.new_empty(shape), x.new_empty(shape, dtype=torch.float32)

    @torch.library.register_fake("leaf_amd::backward")
    def _(x, kernel, pool_w, pool_b, alpha, delta, root, ema_w, K, hop, grad_out, pooled_raw, need_dx, flags):
        F = kernel.shape[0]
        pc = F if alpha is not None else 0
        e = lambda *s: kernel.new_empty(s)
        return [torch.empty_like(kernel), torch.empty_like(pool_w), torch.empty_like(pool_b), e(pc), e(pc), e(pc), e(pc),
                torch.empty_like(x) if need_dx else e(0)]

    def setup_context(ctx, inputs, output):
        x, kernel, pool_w, pool_b, alpha, delta, root, ema_w, K, hop, algo = inputs
        _, raw = output
        ctx.pcen = alpha is not None
        ctx.geom = (K, hop)
        ctx.full = bool(algo & _native.ALGO_FULL_TRANSFORMS)     # Leaf.full_transforms(): no band tasks in the backward either
        ctx.strict = bool(algo & _native.ALGO_STRICT_BAND_CLASSES)
        ctx.save_for_backward(x, kernel, pool_w, pool_b, raw, *([alpha, delta, root, ema_w] if ctx.pcen else []))

    def backward(ctx, grad_out, grad_raw):
        K, hop = ctx.geom
        saved = ctx.saved_tensors
        x, kernel, pool_w, pool_b, raw = saved[:5]
        alpha, delta, root, ema_w = saved[5:] if ctx.pcen else (None,) * 4
        need_dx = ctx.needs_input_grad[0]
        gk, gpw, gpb, ga, gd, gr, gw, gx = torch.ops.leaf_amd.backward(x, kernel, pool_w, pool_b, alpha, delta, root, ema_w, K, hop,
                                                                     grad_out.contiguous(), raw, need_dx,
                                                                     (_native.FLAG_BWD_FULL_TRANSFORMS if ctx.full else 0) |
                                                                     (_native.FLAG_BWD_STRICT_BAND_CLASSES if ctx.strict else 0))
        pc = (ga, gd, gr, gw) if ctx.pcen else (None,) * 4
        return (gx if need_dx else None, gk, gpw, gpb, *pc, None, None, None)

    torch.library.register_autograd("leaf_amd::forward_train", backward, setup_context=setup_context)
    from . import _second_order
    _second_order.register()                  # gradients of gradients: leaf_amd::backward's own autograd formula


def forward(x, kernel, pool_w, pool_b, alpha, delta, root, ema_w, K: int, hop: int, log1p: bool = False,
            algo: int = _native.ALGO_AUTO) -> torch.Tensor:
    return torch.ops.leaf_amd.forward(x, kernel, pool_w, pool_b, alpha, delta, root, ema_w, K, hop, log1p, algo)


def forward_train(x, kernel, pool_w, pool_b, alpha, delta, root, ema_w, K: int, hop: int,
                  algo: int = _native.ALGO_AUTO) -> torch.Tensor:
    return torch.ops.leaf_amd.forward_train(x, kernel, pool_w, pool_b, alpha, delta, root, ema_w, K, hop, algo)[0]
