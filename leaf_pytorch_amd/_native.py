"""ctypes binding of the C ABI in include/leaf_hip.h (libleaf_hip.so, HIP kernels for gfx950).

There is deliberately NO fallback: if the shared library is missing or a call returns a non-zero
status, a RuntimeError is raised.  PyTorch is used only for device memory (tensors own the HBM
buffers, the caching allocator provides the scratch workspace) and for the current HIP stream.
"""
from __future__ import annotations

import ctypes
import os
import subprocess
import threading
from typing import Optional

import torch

_PKG_DIR = os.path.dirname(os.path.abspath(__file__))
_REPO_DIR = os.path.dirname(_PKG_DIR)
LIB_PATH = os.path.join(_PKG_DIR, "libleaf_hip.so")
SRC_PATH = os.path.join(_PKG_DIR, "csrc", "leaf_kernels.hip")
INCLUDE_DIR = os.path.join(_REPO_DIR, "include")

ABI_VERSION = 5
ALGO_AUTO, ALGO_STAGED, ALGO_MFMA, ALGO_FFT, ALGO_FFT_WG, ALGO_FFT_SMALL = 0, 1, 2, 3, 4, 5


def algo_reserve_cus(k: int) -> int:
    """LEAF_ALGO_RESERVE_CUS(k): OR into ``algo`` so that the call leaves ``k`` CUs free for kernels of other streams."""
    return (int(k) & 0xff) << 16


FLAG_PCEN, FLAG_LOG1P, FLAG_IO_BF16, FLAG_BWD_STAGED, FLAG_BWD_MFMA, FLAG_PEAKNORM, FLAG_BWD_FULL_TRANSFORMS = 0x1, 0x2, 0x4, 0x8, 0x10, 0x20, 0x40
FLAG_BWD_STRICT_BAND_CLASSES = 0x80   # leaf_backward_f32: the backward's band classes by round 5's rule alone (default: the forward's bias-aware decision)
ALGO_STREAM_FINALIZE = 1 << 25   # LEAF_ALGO_STREAM_FINALIZE: per-frame sums in an LDS ring, finalized as the blocks complete
ALGO_FULL_TRANSFORMS = 1 << 26   # LEAF_ALGO_FULL_TRANSFORMS: no band-limited filter tasks (every filter on 2048-point transforms)
ALGO_STRICT_BAND_CLASSES = 1 << 27   # LEAF_ALGO_STRICT_BAND_CLASSES: the band classes' energy bound does not follow the pooling bias (round 5's decision)
OPT_PEAKNORM = 1 << 24          # torch.ops.leaf_amd.forward: option bit in `algo` that sets LEAF_FLAG_PEAKNORM (torch_binding.cpp)
STAGE_GABOR_CONV, STAGE_LOWPASS, STAGE_EMA, STAGE_PCEN = 1, 2, 3, 4

_lock = threading.Lock()
_lib: Optional[ctypes.CDLL] = None

_f32p = ctypes.c_void_p
_SIGNATURES = {
    # name: (restype, argtypes)   -- must list every symbol include/leaf_hip.h declares
    "leaf_abi_version": (ctypes.c_int, []),
    "leaf_status_string": (ctypes.c_char_p, [ctypes.c_int]),
    "leaf_num_frames": (ctypes.c_int, [ctypes.c_int] * 3),
    "leaf_workspace_bytes": (ctypes.c_size_t, [ctypes.c_int] * 6),
    "leaf_forward_f32": (ctypes.c_int, [_f32p, ctypes.c_int, ctypes.c_int] + [_f32p] * 7 + [ctypes.c_int] * 5
                         + [_f32p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p]),
    "leaf_auto_algo": (ctypes.c_int, [ctypes.c_int] * 5),
    "leaf_fft_plan_info": (ctypes.c_int, [ctypes.c_int] * 5 + [ctypes.POINTER(ctypes.c_int)]),
    "leaf_band_classes_f32": (ctypes.c_int, [_f32p, _f32p, _f32p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p,
                                             ctypes.c_size_t, ctypes.c_void_p]),
    "leaf_forward_profiled_f32": (ctypes.c_int, [_f32p, ctypes.c_int, ctypes.c_int] + [_f32p] * 7 + [ctypes.c_int] * 5
                                  + [_f32p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p,
                                     ctypes.POINTER(ctypes.c_float)]),
    "leaf_backward_workspace_bytes": (ctypes.c_size_t, [ctypes.c_int] * 7),
    "leaf_backward_f32": (ctypes.c_int, [_f32p, ctypes.c_int, ctypes.c_int] + [_f32p] * 7 + [ctypes.c_int] * 4 + [_f32p] * 10
                          + [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p]),
    "leaf_forward_save_f32": (ctypes.c_int, [_f32p, ctypes.c_int, ctypes.c_int] + [_f32p] * 7 + [ctypes.c_int] * 5
                              + [_f32p, _f32p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p]),
    "leaf_gabor_taps_f32": (ctypes.c_int, [_f32p, ctypes.c_int, ctypes.c_int, _f32p, ctypes.c_void_p]),
    "leaf_lowpass_window_f32": (ctypes.c_int, [_f32p, ctypes.c_int, ctypes.c_int, _f32p, ctypes.c_void_p]),
    "leaf_gabor_conv_f32": (ctypes.c_int, [_f32p, ctypes.c_int, ctypes.c_int, _f32p, ctypes.c_int, ctypes.c_int,
                                           _f32p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p]),
    "leaf_squared_modulus_f32": (ctypes.c_int, [_f32p, ctypes.c_int, ctypes.c_int, ctypes.c_int, _f32p,
                                                ctypes.c_void_p]),
    "leaf_gaussian_lowpass_f32": (ctypes.c_int, [_f32p, ctypes.c_int, ctypes.c_int, ctypes.c_int, _f32p, _f32p,
                                                 ctypes.c_int, ctypes.c_int, _f32p, ctypes.c_void_p,
                                                 ctypes.c_size_t, ctypes.c_void_p]),
    "leaf_ema_f32": (ctypes.c_int, [_f32p, ctypes.c_int, ctypes.c_int, ctypes.c_int, _f32p, _f32p, ctypes.c_void_p]),
    "leaf_pcen_f32": (ctypes.c_int, [_f32p, ctypes.c_int, ctypes.c_int, ctypes.c_int, _f32p, _f32p, _f32p, _f32p,
                                     ctypes.c_float, _f32p, ctypes.c_void_p]),
    "leaf_pcen_stream_f32": (ctypes.c_int, [_f32p, ctypes.c_int, ctypes.c_int, ctypes.c_int, _f32p, _f32p, _f32p, _f32p,
                                            ctypes.c_float, ctypes.c_int, _f32p, _f32p, _f32p, ctypes.c_void_p]),
    "leaf_stage_backward_workspace_bytes": (ctypes.c_size_t, [ctypes.c_int] * 6),
    "leaf_gabor_conv_backward_f32": (ctypes.c_int, [_f32p, ctypes.c_int, ctypes.c_int, _f32p, ctypes.c_int, ctypes.c_int,
                                                    _f32p, _f32p, _f32p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p]),
    "leaf_squared_modulus_backward_f32": (ctypes.c_int, [_f32p, _f32p, ctypes.c_int, ctypes.c_int, ctypes.c_int, _f32p,
                                                         ctypes.c_void_p]),
    "leaf_gaussian_lowpass_backward_f32": (ctypes.c_int, [_f32p, _f32p, ctypes.c_int, ctypes.c_int, ctypes.c_int, _f32p,
                                                          ctypes.c_int, ctypes.c_int, _f32p, _f32p, _f32p, ctypes.c_void_p,
                                                          ctypes.c_size_t, ctypes.c_void_p]),
    "leaf_ema_backward_f32": (ctypes.c_int, [_f32p, _f32p, ctypes.c_int, ctypes.c_int, ctypes.c_int, _f32p, _f32p, _f32p,
                                             ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p]),
    "leaf_pcen_backward_f32": (ctypes.c_int, [_f32p, _f32p, ctypes.c_int, ctypes.c_int, ctypes.c_int] + [_f32p] * 4
                               + [ctypes.c_float] + [_f32p] * 5 + [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p]),
    "leaf_peak_normalize_f32": (ctypes.c_int, [_f32p, ctypes.c_int, ctypes.c_int, _f32p, ctypes.c_void_p]),
    "leaf_fft_tables_bytes": (ctypes.c_size_t, [ctypes.c_int] * 3),
    "leaf_fft_prepare_tables_f32": (ctypes.c_int, [_f32p, _f32p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p,
                                                   ctypes.c_size_t, ctypes.c_void_p]),
    "leaf_forward_prepared_f32": (ctypes.c_int, [_f32p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_size_t]
                                  + [_f32p] * 5 + [ctypes.c_int] * 4
                                  + [_f32p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p]),
}
EXPORTED_SYMBOLS = tuple(_SIGNATURES)


def _translation_units(csrc: str):
    """leaf_kernels.hip (C ABI, host logic, small kernels) + one inst_*.hip per family of big kernel templates."""
    return [SRC_PATH] + sorted(os.path.join(csrc, f) for f in os.listdir(csrc) if f.startswith("inst_") and f.endswith(".hip"))


def _includes_of(path: str, csrc: str, seen=None) -> set:
    """Transitive closure of the quoted #include files of one translation unit (csrc/ and include/ only)."""
    seen = set() if seen is None else seen
    with open(path) as fh:
        for line in fh:
            line = line.strip()
            if line.startswith('#include "'):
                name = line.split('"')[1]
                for base in (csrc, INCLUDE_DIR):
                    cand = os.path.join(base, name)
                    if os.path.exists(cand) and cand not in seen:
                        seen.add(cand)
                        _includes_of(cand, csrc, seen)
    return seen


def build(force: bool = False, verbose: bool = False, jobs: Optional[int] = None, variant: Optional[str] = None,
          extra_flags: Optional[str] = None) -> str:
    """Compile csrc/*.hip for gfx950 into libleaf_hip.so (in-tree).  Needs hipcc, not a GPU.

    The translation units are compiled in parallel into build/*.o (git-ignored) and only those whose sources or headers
    changed are recompiled; the shared library is linked from the objects.  ``variant`` (tools only) builds a second
    library with ``extra_flags`` into build/variants/<variant>/ and returns its path; the product library is untouched."""
    csrc = os.path.dirname(SRC_PATH)
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    extra = (os.environ.get("LEAF_HIPCC_EXTRA", "") if extra_flags is None else extra_flags).split()
    LIB_PATH = globals()["LIB_PATH"] if variant is None else os.path.join(_PKG_DIR, "build", "variants", variant, "libleaf_hip.so")
    # -fno-slp-vectorize: the SLP vectorizer packs the FFT butterflies into v_pk_*_f32 (no faster than two scalar ops on
    # gfx950, tools/ubench_valu.hip) at the price of hundreds of v_mov shuffles and ~35 extra VGPRs per kernel
    flags = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fno-slp-vectorize", "-fPIC", "-I", INCLUDE_DIR] + extra
    obj_dir = os.path.join(_PKG_DIR, "build") if variant is None else os.path.dirname(LIB_PATH)
    os.makedirs(obj_dir, exist_ok=True)
    stamp = os.path.join(obj_dir, "flags.txt")
    flag_text = " ".join([hipcc] + flags)
    if not os.path.exists(stamp) or open(stamp).read() != flag_text:
        force = True                                         # different flags (LEAF_HIPCC_EXTRA): every object is stale
    units, objs, todo = _translation_units(csrc), [], []
    for src in units:
        obj = os.path.join(obj_dir, os.path.splitext(os.path.basename(src))[0] + ".o")
        objs.append(obj)
        deps = [src] + sorted(_includes_of(src, csrc))
        if force or not os.path.exists(obj) or os.path.getmtime(obj) < max(os.path.getmtime(d) for d in deps):
            todo.append((src, obj))
    if not todo and os.path.exists(LIB_PATH) and os.path.getmtime(LIB_PATH) >= max(os.path.getmtime(o) for o in objs):
        return LIB_PATH
    jobs = jobs or int(os.environ.get("LEAF_BUILD_JOBS", "0")) or min(len(todo) or 1, os.cpu_count() or 1)
    procs, failed = [], []
    pending = list(todo)
    while pending or procs:
        while pending and len(procs) < jobs:
            src, obj = pending.pop(0)
            cmd = [hipcc] + flags + ["-c", src, "-o", obj + ".tmp"]
            if verbose:
                print(" ".join(cmd), flush=True)
            procs.append((subprocess.Popen(cmd), src, obj))
        proc, src, obj = procs.pop(0)
        if proc.wait() != 0:
            failed.append(src)
        else:
            os.replace(obj + ".tmp", obj)
    if failed:
        raise subprocess.CalledProcessError(1, f"hipcc failed for {failed}")
    with open(stamp, "w") as fh:
        fh.write(flag_text)
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC"] + objs + ["-o", LIB_PATH + ".tmp"]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.run(cmd, check=True)
    os.replace(LIB_PATH + ".tmp", LIB_PATH)
    return LIB_PATH


def load() -> ctypes.CDLL:
    """dlopen libleaf_hip.so and attach prototypes; raises RuntimeError (never falls back) if absent."""
    global _lib
    if _lib is not None:
        return _lib
    with _lock:
        if _lib is not None:
            return _lib
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"{LIB_PATH} not found: the HIP extension is required (no CPU/eager fallback exists). "
                "Build it with `python -c 'import __graft_entry__ as g; g.build()'` or leaf_pytorch_amd.build().")
        lib = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in _SIGNATURES.items():
            fn = getattr(lib, name)          # AttributeError here = ABI mismatch, surfaced loudly
            fn.restype, fn.argtypes = res, args
        if lib.leaf_abi_version() != ABI_VERSION:
            raise RuntimeError("libleaf_hip.so ABI version mismatch")
        _lib = lib
    return _lib


def check(status: int, what: str) -> None:
    if status != 0:
        msg = load().leaf_status_string(status).decode()
        raise RuntimeError(f"{what} failed: {msg} (status {status})")


def _ptr(t: Optional[torch.Tensor]):
    return None if t is None else ctypes.c_void_p(t.data_ptr())


def _dev_f32(t: torch.Tensor, name: str, device: torch.device) -> torch.Tensor:
    if t.device != device:
        raise RuntimeError(f"{name} is on {t.device}, expected {device}")
    if t.dtype != torch.float32:
        raise RuntimeError(f"{name} must be float32, got {t.dtype}")
    return t.detach().contiguous()


def require_hip(x: torch.Tensor, who: str) -> None:
    if x.device.type != "cuda":
        raise RuntimeError(
            f"{who}: input is on '{x.device}'. leaf_pytorch_amd runs only on an AMD GPU through its HIP kernels; "
            "there is no CPU path in the product (the CPU restatement lives in oracle/ for tests only).")


def stream_ptr(device: torch.device) -> ctypes.c_void_p:
    return ctypes.c_void_p(torch.cuda.current_stream(device).cuda_stream)


def num_frames(T: int, K: int, hop: int) -> int:
    return load().leaf_num_frames(T, K, hop)


def fft_plan_info(B: int, T: int, F: int, K: int, hop: int) -> Optional[dict]:
    """Plan of the overlap-save path (leaf_fft_plan_info), or None when it does not cover the geometry."""
    info = (ctypes.c_int * 8)()
    if load().leaf_fft_plan_info(B, T, F, K, hop, info) != 0:
        return None
    keys = ("fft_n", "block_len", "blocks_per_clip", "filters_per_task", "filter_groups", "slots", "row_buffers", "lds_bytes")
    return dict(zip(keys, (int(v) for v in info)))


def band_classes(kernel: torch.Tensor, pool_w: torch.Tensor, K: int, hop: int,
                 pool_b: Optional[torch.Tensor] = None) -> Optional[torch.Tensor]:
    """Inverse-transform length (256 / 512 / 2048; 512 / 4096 on the 4096-sample plan of the 32 kHz window) each filter gets from the band-limited filter tasks for these parameters
    (leaf_band_classes_f32), as an int32 tensor [F] on the parameters' device; None for a geometry without band tasks.
    ``pool_b``: the pooling biases the decision is taken for (what a forward call with them runs: the energy bound follows the
    bias, ABI 5); None: the strict decision (LEAF_ALGO_STRICT_BAND_CLASSES; the backward's)."""
    lib = load()
    require_hip(kernel, "band_classes")
    dev = kernel.device
    kernel = _dev_f32(kernel, "kernel", dev)
    pool_w = _dev_f32(pool_w.reshape(-1), "pool_w", dev)
    pool_b = None if pool_b is None else _dev_f32(pool_b.reshape(-1), "pool_b", dev)
    F = kernel.shape[0]
    out = torch.empty(F, dtype=torch.int32, device=dev)
    with torch.cuda.device(dev):
        ws = workspace(max(lib.leaf_fft_tables_bytes(F, K, hop), lib.leaf_workspace_bytes(1, 8192, F, K, hop, ALGO_FFT_WG)), dev)
        rc = lib.leaf_band_classes_f32(_ptr(kernel), _ptr(pool_w), _ptr(pool_b), F, K, hop, _ptr(out), _ptr(ws), ws.numel(), stream_ptr(dev))
    if rc == -8:
        return None
    check(rc, "leaf_band_classes_f32")
    return out


def workspace(nbytes: int, device: torch.device) -> torch.Tensor:
    return torch.empty(max(nbytes, 4), dtype=torch.uint8, device=device)


def batch_slices(B: int, T: int):
    """Slices of whole clips for a batch beyond one C-ABI call.  The C ABI indexes the samples of one call with 32 bits and
    refuses B * T >= 2^31 (LEAF_ERR_BAD_SHAPE); the reference's conv1d takes any batch (frontend.py:78-89), and clips are
    independent, so such a batch goes through in as few balanced slices as possible (the same plan as csrc/torch_binding.cpp)."""
    most = max(1, ((1 << 31) - 1) // max(T, 1))
    calls = max(1, -(-B // most))
    per = -(-B // calls)
    return [(b0, min(B, b0 + per)) for b0 in range(0, B, per)]


def leaf_forward(x: torch.Tensor, kernel, pool_w, pool_b, alpha, delta, root, ema_w, K: int, hop: int,
                 pcen: bool = True, log1p: bool = False, algo: int = ALGO_AUTO,
                 out: Optional[torch.Tensor] = None, save_raw: bool = False, peak_normalize: bool = False):
    """x (B,1,T) or (B,T) float32 on a HIP device -> (B,F,T').  Wraps leaf_forward_f32 (leaf_forward_save_f32 when
    ``save_raw``: then returns (out, pooled_raw) for the backward)."""
    lib = load()
    require_hip(x, "leaf_forward")
    if x.dim() == 3:
        if x.shape[1] != 1:
            raise RuntimeError(f"expected input of shape (B,1,T), got {tuple(x.shape)}")
        x2 = x[:, 0, :]
    elif x.dim() == 2:
        x2 = x
    else:
        raise RuntimeError(f"expected input of shape (B,1,T), got {tuple(x.shape)}")
    dev = x.device
    io_bf16 = x2.dtype == torch.bfloat16           # extension: bf16 waveform in, bf16 features out, fp32 arithmetic
    x2 = x2.detach().contiguous() if io_bf16 else _dev_f32(x2, "x", dev)
    B, T = x2.shape
    F = kernel.shape[0]
    kernel = _dev_f32(kernel, "kernel", dev)
    pool_w = _dev_f32(pool_w.reshape(-1), "pool_w", dev)
    pool_b = _dev_f32(pool_b, "pool_b", dev)
    flags = FLAG_IO_BF16 if io_bf16 else 0
    if peak_normalize:
        flags |= FLAG_PEAKNORM                     # forward of the peak-normalised clips, the scale folded into the finalize
    if pcen:
        flags |= FLAG_PCEN
        alpha, delta, root, ema_w = (_dev_f32(t, n, dev) for t, n in
                                     ((alpha, "alpha"), (delta, "delta"), (root, "root"), (ema_w, "ema_w")))
    else:
        alpha = delta = root = ema_w = None
        if log1p:
            flags |= FLAG_LOG1P
    TP = lib.leaf_num_frames(T, K, hop)
    if TP < 1:
        raise RuntimeError(f"bad shape B={B} T={T} K={K} hop={hop}")
    if out is not None and (out.dtype != (torch.bfloat16 if io_bf16 else torch.float32) or not out.is_contiguous()
                            or tuple(out.shape) != (B, F, TP) or out.device != dev):
        raise RuntimeError(f"out must be a contiguous {(B, F, TP)} tensor on {dev} matching the input dtype (float32 or bfloat16)")
    if (algo & 0xff) not in (ALGO_AUTO, ALGO_STAGED, ALGO_MFMA, ALGO_FFT, ALGO_FFT_WG, ALGO_FFT_SMALL):
        raise RuntimeError(f"unknown algorithm selector {algo & 0xff}")
    if B == 0:
        # the empty batch: (0, F, T') like the reference (frontend.py:78-89 -> convolution.py:97); nothing is launched
        # (`out`, the selector and the flags are validated above, and a caller-supplied `out` is what comes back)
        if save_raw and peak_normalize:
            raise RuntimeError("the folded PeakNormalization prologue is forward-only")
        empty = out if out is not None else torch.empty((0, F, TP), dtype=torch.bfloat16 if io_bf16 else torch.float32, device=dev)
        return (empty, torch.empty((0, F, TP), dtype=torch.float32, device=dev)) if save_raw else empty
    if out is None:
        out = torch.empty((B, F, TP), dtype=torch.bfloat16 if io_bf16 else torch.float32, device=dev)
    if B * T >= (1 << 31):
        # one C-ABI call per slice of whole clips, into the one output (see batch_slices)
        raw = torch.empty((B, F, TP), dtype=torch.float32, device=dev) if save_raw else None
        for b0, b1 in batch_slices(B, T):
            r = leaf_forward(x2[b0:b1], kernel, pool_w, pool_b, alpha, delta, root, ema_w, K, hop, pcen=pcen, log1p=log1p, algo=algo,
                             out=out[b0:b1], save_raw=save_raw, peak_normalize=peak_normalize)
            if save_raw:
                raw[b0:b1].copy_(r[1])
        return (out, raw) if save_raw else out
    with torch.cuda.device(dev):
        nbytes = lib.leaf_workspace_bytes(B, T, F, K, hop, algo)
        ws = workspace(nbytes, dev)
        if save_raw:
            raw = torch.empty((B, F, TP), dtype=torch.float32, device=dev)
            rc = lib.leaf_forward_save_f32(_ptr(x2), B, T, _ptr(kernel), _ptr(pool_w), _ptr(pool_b), _ptr(alpha),
                                           _ptr(delta), _ptr(root), _ptr(ema_w), F, K, hop, flags, algo, _ptr(out),
                                           _ptr(raw), _ptr(ws), ws.numel(), stream_ptr(dev))
            check(rc, "leaf_forward_save_f32")
            return out, raw
        rc = lib.leaf_forward_f32(_ptr(x2), B, T, _ptr(kernel), _ptr(pool_w), _ptr(pool_b), _ptr(alpha), _ptr(delta),
                                  _ptr(root), _ptr(ema_w), F, K, hop, flags, algo, _ptr(out), _ptr(ws),
                                  ws.numel(), stream_ptr(dev))
    check(rc, "leaf_forward_f32")
    return out


def leaf_backward(x, kernel, pool_w, pool_b, alpha, delta, root, ema_w, K: int, hop: int, grad_out: torch.Tensor,
                  pcen: bool = True, need_dx: bool = False, staged: bool = False,
                  pooled_raw: Optional[torch.Tensor] = None, mfma: bool = False, full_transforms: bool = False,
                  strict_band_classes: bool = False):
    """Gradients of the forward w.r.t. (kernel, pool_w, pool_b, alpha, delta, root, ema_w[, x]).  Wraps leaf_backward_f32.
    ``staged`` / ``mfma`` force the staged kernels / the fused MFMA backward (default: the overlap-save backward where
    it applies, else MFMA, else staged)."""
    lib = load()
    require_hip(x, "leaf_backward")
    dev = x.device
    x2 = _dev_f32(x[:, 0, :] if x.dim() == 3 else x, "x", dev)
    B, T = x2.shape
    F = kernel.shape[0]
    kernel = _dev_f32(kernel, "kernel", dev)
    pw = _dev_f32(pool_w.reshape(-1), "pool_w", dev)
    pb = _dev_f32(pool_b, "pool_b", dev)
    go = _dev_f32(grad_out, "grad_out", dev)
    TP = lib.leaf_num_frames(T, K, hop)
    if tuple(go.shape) != (B, F, TP):
        raise RuntimeError(f"grad_out has shape {tuple(go.shape)}, expected {(B, F, TP)}")
    g_kernel = torch.empty_like(kernel)
    g_pw, g_pb = torch.empty_like(pw), torch.empty_like(pb)
    if pcen:
        alpha, delta, root, ema_w = (_dev_f32(t, "pcen param", dev) for t in (alpha, delta, root, ema_w))
        g_pc = [torch.empty(F, dtype=torch.float32, device=dev) for _ in range(4)]
    else:
        alpha = delta = root = ema_w = None
        g_pc = [None] * 4
    g_x = torch.empty_like(x2) if need_dx else None
    if B == 0:                                     # the sum over no clips: zero parameter gradients, nothing launched
        for g in (g_kernel, g_pw, g_pb, *g_pc):
            if g is not None:
                g.zero_()
        return g_kernel, g_pw.reshape(pool_w.shape), g_pb, g_pc[0], g_pc[1], g_pc[2], g_pc[3], g_x
    if B * T >= (1 << 31):
        # slices of whole clips (batch_slices); parameter gradients added in slice order (fixed: bit-reproducible)
        total = None
        for b0, b1 in batch_slices(B, T):
            g = leaf_backward(x2[b0:b1], kernel, pool_w, pool_b, alpha, delta, root, ema_w, K, hop, go[b0:b1], pcen=pcen, need_dx=need_dx,
                              staged=staged, pooled_raw=None if pooled_raw is None else pooled_raw[b0:b1], mfma=mfma,
                              full_transforms=full_transforms, strict_band_classes=strict_band_classes)
            if need_dx:
                g_x[b0:b1].copy_(g[7])
            total = list(g[:7]) if total is None else [None if a is None else a.add_(b) for a, b in zip(total, g[:7])]
        return (*total, g_x)
    flags = ((FLAG_PCEN if pcen else 0) | (FLAG_BWD_STAGED if staged else 0) | (FLAG_BWD_MFMA if mfma else 0) |
             (FLAG_BWD_FULL_TRANSFORMS if full_transforms else 0) |   # full_transforms: no band-limited filter tasks in the backward
             (FLAG_BWD_STRICT_BAND_CLASSES if strict_band_classes else 0))
    with torch.cuda.device(dev):
        # sized for the path these flags select (a few MB for the overlap-save backward, not the staged path's dL/dy)
        ws = workspace(lib.leaf_backward_workspace_bytes(B, T, F, K, hop, flags, int(need_dx)), dev)
        rc = lib.leaf_backward_f32(_ptr(x2), B, T, _ptr(kernel), _ptr(pw), _ptr(pb), _ptr(alpha), _ptr(delta), _ptr(root),
                                   _ptr(ema_w), F, K, hop, flags,
                                   _ptr(go), _ptr(pooled_raw), _ptr(g_kernel), _ptr(g_pw),
                                   _ptr(g_pb), _ptr(g_pc[0]), _ptr(g_pc[1]), _ptr(g_pc[2]), _ptr(g_pc[3]), _ptr(g_x),
                                   _ptr(ws), ws.numel(), stream_ptr(dev))
    check(rc, "leaf_backward_f32")
    return g_kernel, g_pw.reshape(pool_w.shape), g_pb, g_pc[0], g_pc[1], g_pc[2], g_pc[3], g_x


def leaf_forward_profiled(x, kernel, pool_w, pool_b, alpha, delta, root, ema_w, K: int, hop: int, pcen: bool = True,
                          algo: int = ALGO_AUTO, log1p: bool = False):
    """Measurement call: returns (out, [taps_ms, fused_ms, finalize_ms]) from HIP events on the current stream.  Same flags
    as ``leaf_forward`` (PCEN on / off, log1p, bfloat16 I/O when ``x`` is bfloat16), so every BASELINE config can be timed."""
    lib = load()
    require_hip(x, "leaf_forward_profiled")
    dev = x.device
    x2 = x[:, 0, :] if x.dim() == 3 else x
    io_bf16 = x2.dtype == torch.bfloat16
    x2 = x2.detach().contiguous() if io_bf16 else _dev_f32(x2, "x", dev)
    B, T = x2.shape
    F = kernel.shape[0]
    kernel = _dev_f32(kernel, "kernel", dev)
    pool_w = _dev_f32(pool_w.reshape(-1), "pool_w", dev)
    pool_b = _dev_f32(pool_b, "pool_b", dev)
    flags = FLAG_IO_BF16 if io_bf16 else 0
    if pcen:
        flags |= FLAG_PCEN
        alpha, delta, root, ema_w = (_dev_f32(t, "pcen param", dev) for t in (alpha, delta, root, ema_w))
    else:
        alpha = delta = root = ema_w = None
        if log1p:
            flags |= FLAG_LOG1P
    out = torch.empty((B, F, lib.leaf_num_frames(T, K, hop)), dtype=torch.bfloat16 if io_bf16 else torch.float32, device=dev)
    ms = (ctypes.c_float * 3)()
    with torch.cuda.device(dev):
        ws = workspace(lib.leaf_workspace_bytes(B, T, F, K, hop, algo), dev)
        rc = lib.leaf_forward_profiled_f32(_ptr(x2), B, T, _ptr(kernel), _ptr(pool_w), _ptr(pool_b), _ptr(alpha),
                                           _ptr(delta), _ptr(root), _ptr(ema_w), F, K, hop, flags, algo,
                                           _ptr(out), _ptr(ws), ws.numel(), stream_ptr(dev), ms)
    check(rc, "leaf_forward_profiled_f32")
    return out, [float(v) for v in ms]


def gabor_taps(kernel: torch.Tensor, K: int) -> torch.Tensor:
    lib = load(); require_hip(kernel, "gabor_taps")
    kernel = _dev_f32(kernel, "kernel", kernel.device)
    F = kernel.shape[0]
    taps = torch.empty((2 * F, K), dtype=torch.float32, device=kernel.device)
    with torch.cuda.device(kernel.device):
        check(lib.leaf_gabor_taps_f32(_ptr(kernel), F, K, _ptr(taps), stream_ptr(kernel.device)), "leaf_gabor_taps_f32")
    return taps


def lowpass_window(pool_w: torch.Tensor, K: int) -> torch.Tensor:
    lib = load(); require_hip(pool_w, "lowpass_window")
    w = _dev_f32(pool_w.reshape(-1), "pool_w", pool_w.device)
    g = torch.empty((w.numel(), K), dtype=torch.float32, device=w.device)
    with torch.cuda.device(w.device):
        check(lib.leaf_lowpass_window_f32(_ptr(w), w.numel(), K, _ptr(g), stream_ptr(w.device)), "leaf_lowpass_window_f32")
    return g


def gabor_conv(x: torch.Tensor, kernel: torch.Tensor, K: int) -> torch.Tensor:
    lib = load(); require_hip(x, "gabor_conv")
    if x.dim() != 3 or x.shape[1] != 1:
        raise RuntimeError(f"expected input of shape (B,1,T), got {tuple(x.shape)}")
    dev = x.device
    x2 = _dev_f32(x[:, 0, :], "x", dev)
    kernel = _dev_f32(kernel, "kernel", dev)
    B, T = x2.shape; F = kernel.shape[0]
    y = torch.empty((B, 2 * F, T), dtype=torch.float32, device=dev)
    if B == 0:                                     # the empty batch passes through every stage as an empty tensor
        return y
    with torch.cuda.device(dev):
        ws = workspace(2 * F * K * 4, dev)
        check(lib.leaf_gabor_conv_f32(_ptr(x2), B, T, _ptr(kernel), F, K, _ptr(y), _ptr(ws), ws.numel(), stream_ptr(dev)),
              "leaf_gabor_conv_f32")
    return y


def squared_modulus(y: torch.Tensor) -> torch.Tensor:
    lib = load(); require_hip(y, "squared_modulus")
    y = _dev_f32(y, "y", y.device)
    B, C2, T = y.shape
    if C2 % 2:
        raise RuntimeError("channel count must be even (interleaved re/im)")
    e = torch.empty((B, C2 // 2, T), dtype=torch.float32, device=y.device)
    if B == 0:
        return e
    with torch.cuda.device(y.device):
        check(lib.leaf_squared_modulus_f32(_ptr(y), B, C2 // 2, T, _ptr(e), stream_ptr(y.device)), "leaf_squared_modulus_f32")
    return e


def gaussian_lowpass(e: torch.Tensor, pool_w: torch.Tensor, pool_b: Optional[torch.Tensor], K: int, hop: int) -> torch.Tensor:
    lib = load(); require_hip(e, "gaussian_lowpass")
    dev = e.device
    e = _dev_f32(e, "e", dev)
    B, F, T = e.shape
    w = _dev_f32(pool_w.reshape(-1), "pool_w", dev)
    b = None if pool_b is None else _dev_f32(pool_b, "pool_b", dev)
    TP = lib.leaf_num_frames(T, K, hop)
    pooled = torch.empty((B, F, TP), dtype=torch.float32, device=dev)
    if B == 0:
        return pooled
    with torch.cuda.device(dev):
        ws = workspace(F * K * 4, dev)
        check(lib.leaf_gaussian_lowpass_f32(_ptr(e), B, F, T, _ptr(w), _ptr(b), K, hop, _ptr(pooled), _ptr(ws), ws.numel(),
                                            stream_ptr(dev)), "leaf_gaussian_lowpass_f32")
    return pooled


def ema(p: torch.Tensor, ema_w: torch.Tensor) -> torch.Tensor:
    lib = load(); require_hip(p, "ema")
    dev = p.device
    p = _dev_f32(p, "p", dev); B, F, TP = p.shape
    w = _dev_f32(ema_w.reshape(-1).expand(F) if ema_w.numel() == 1 else ema_w, "ema_w", dev)
    out = torch.empty_like(p)
    if B == 0:
        return out
    with torch.cuda.device(dev):
        check(lib.leaf_ema_f32(_ptr(p), B, F, TP, _ptr(w), _ptr(out), stream_ptr(dev)), "leaf_ema_f32")
    return out


def pcen(p: torch.Tensor, alpha, delta, root, ema_w, floor: float) -> torch.Tensor:
    lib = load(); require_hip(p, "pcen")
    dev = p.device
    p = _dev_f32(p, "p", dev); B, F, TP = p.shape
    alpha, delta, root = (_dev_f32(t, n, dev) for t, n in ((alpha, "alpha"), (delta, "delta"), (root, "root")))
    w = _dev_f32(ema_w.reshape(-1).expand(F) if ema_w.numel() == 1 else ema_w, "ema_w", dev)
    out = torch.empty_like(p)
    if B == 0:
        return out
    with torch.cuda.device(dev):
        check(lib.leaf_pcen_f32(_ptr(p), B, F, TP, _ptr(alpha), _ptr(delta), _ptr(root), _ptr(w), float(floor), _ptr(out),
                                stream_ptr(dev)), "leaf_pcen_f32")
    return out


# ---- stage backwards (what autograd derives for a sub-module called on its own; modules.py wraps them) ----------

def pcen_stream(p: torch.Tensor, alpha, delta, root, ema_w, floor: float, ema_state: Optional[torch.Tensor] = None,
                log1p: bool = False):
    """leaf_pcen_stream_f32: PCEN of one chunk (B,F,n) of floored pooled frames with the smoother state carried between
    calls.  Returns (out, new_state); ``alpha is None``: no PCEN (state stays None)."""
    lib = load()
    require_hip(p, "pcen_stream")
    dev = p.device
    p = _dev_f32(p, "p", dev)
    B, F, n = p.shape
    out = torch.empty_like(p)
    if alpha is None:
        with torch.cuda.device(dev):
            check(lib.leaf_pcen_stream_f32(_ptr(p), B, F, n, None, None, None, None, float(floor), int(log1p), None, None, _ptr(out),
                                           stream_ptr(dev)), "leaf_pcen_stream_f32")
        return out, None
    alpha, delta, root, ema_w = (_dev_f32(t, nm, dev) for t, nm in
                                 ((alpha, "alpha"), (delta, "delta"), (root, "root"), (ema_w, "ema_w")))
    new_state = torch.empty((B, F), dtype=torch.float32, device=dev)
    if ema_state is not None:
        ema_state = _dev_f32(ema_state, "ema_state", dev)
    with torch.cuda.device(dev):
        check(lib.leaf_pcen_stream_f32(_ptr(p), B, F, n, _ptr(alpha), _ptr(delta), _ptr(root), _ptr(ema_w), float(floor), 0,
                                       _ptr(ema_state), _ptr(new_state), _ptr(out), stream_ptr(dev)), "leaf_pcen_stream_f32")
    return out, new_state


def _stage_ws(stage: int, B: int, T: int, F: int, K: int, hop: int, dev) -> torch.Tensor:
    return workspace(load().leaf_stage_backward_workspace_bytes(stage, B, T, F, K, hop), dev)


def gabor_conv_backward(x, kernel, K: int, grad_y, need_dk: bool = True, need_dx: bool = False):
    lib = load(); require_hip(x, "gabor_conv_backward")
    dev = x.device
    x2 = _dev_f32(x[:, 0, :], "x", dev)
    kernel = _dev_f32(kernel, "kernel", dev)
    gy = _dev_f32(grad_y, "grad_y", dev)
    B, T = x2.shape; F = kernel.shape[0]
    gk = torch.empty_like(kernel) if need_dk else None
    gx = torch.empty_like(x2) if need_dx else None
    if B == 0:                                     # the sum over no clips
        return (gk.zero_() if gk is not None else None), (gx.reshape(x.shape) if gx is not None else None)
    with torch.cuda.device(dev):
        ws = _stage_ws(STAGE_GABOR_CONV, B, T, F, K, 1, dev)
        check(lib.leaf_gabor_conv_backward_f32(_ptr(x2), B, T, _ptr(kernel), F, K, _ptr(gy), _ptr(gk), _ptr(gx), _ptr(ws),
                                               ws.numel(), stream_ptr(dev)), "leaf_gabor_conv_backward_f32")
    return gk, (gx.reshape(x.shape) if gx is not None else None)


def squared_modulus_backward(y, grad_e):
    lib = load(); require_hip(y, "squared_modulus_backward")
    dev = y.device
    y = _dev_f32(y, "y", dev); ge = _dev_f32(grad_e, "grad_e", dev)
    B, C2, T = y.shape
    gy = torch.empty_like(y)
    if B == 0:
        return gy
    with torch.cuda.device(dev):
        check(lib.leaf_squared_modulus_backward_f32(_ptr(y), _ptr(ge), B, C2 // 2, T, _ptr(gy), stream_ptr(dev)),
              "leaf_squared_modulus_backward_f32")
    return gy


def gaussian_lowpass_backward(e, pool_w, K: int, hop: int, grad_pooled, need_de: bool = True, need_dw: bool = True,
                              need_db: bool = True):
    lib = load(); require_hip(e, "gaussian_lowpass_backward")
    dev = e.device
    e = _dev_f32(e, "e", dev); gp = _dev_f32(grad_pooled, "grad_pooled", dev)
    w = _dev_f32(pool_w.reshape(-1), "pool_w", dev)
    B, F, T = e.shape
    ge = torch.empty_like(e) if need_de else None
    gw = torch.empty_like(w) if need_dw else None
    gb = torch.empty_like(w) if need_db else None
    if B == 0:
        return ge, (gw.zero_().reshape(pool_w.shape) if gw is not None else None), (gb.zero_() if gb is not None else None)
    with torch.cuda.device(dev):
        ws = _stage_ws(STAGE_LOWPASS, B, T, F, K, hop, dev)
        check(lib.leaf_gaussian_lowpass_backward_f32(_ptr(e), _ptr(gp), B, F, T, _ptr(w), K, hop, _ptr(ge), _ptr(gw), _ptr(gb),
                                                     _ptr(ws), ws.numel(), stream_ptr(dev)), "leaf_gaussian_lowpass_backward_f32")
    return ge, (gw.reshape(pool_w.shape) if gw is not None else None), gb


def ema_backward(p, ema_w, grad_ema):
    lib = load(); require_hip(p, "ema_backward")
    dev = p.device
    p = _dev_f32(p, "p", dev); g = _dev_f32(grad_ema, "grad_ema", dev)
    B, F, TP = p.shape
    shared = ema_w.numel() == 1
    w = _dev_f32(ema_w.reshape(-1).expand(F) if shared else ema_w, "ema_w", dev)
    gp, gw = torch.empty_like(p), torch.empty(F, dtype=torch.float32, device=dev)
    if B == 0:
        gw.zero_()
        return gp, (gw.sum().reshape(ema_w.shape) if shared else gw.reshape(ema_w.shape))
    with torch.cuda.device(dev):
        ws = _stage_ws(STAGE_EMA, B, TP, F, 1, 1, dev)
        check(lib.leaf_ema_backward_f32(_ptr(p), _ptr(g), B, F, TP, _ptr(w), _ptr(gp), _ptr(gw), _ptr(ws), ws.numel(),
                                        stream_ptr(dev)), "leaf_ema_backward_f32")
    return gp, (gw.sum().reshape(ema_w.shape) if shared else gw.reshape(ema_w.shape))


def pcen_backward(p, alpha, delta, root, ema_w, floor: float, grad_out):
    lib = load(); require_hip(p, "pcen_backward")
    dev = p.device
    p = _dev_f32(p, "p", dev); g = _dev_f32(grad_out, "grad_out", dev)
    B, F, TP = p.shape
    alpha, delta, root = (_dev_f32(t, n, dev) for t, n in ((alpha, "alpha"), (delta, "delta"), (root, "root")))
    shared = ema_w.numel() == 1
    w = _dev_f32(ema_w.reshape(-1).expand(F) if shared else ema_w, "ema_w", dev)
    gp = torch.empty_like(p)
    ga, gd, gr, gw = (torch.empty(F, dtype=torch.float32, device=dev) for _ in range(4))
    if B == 0:
        for g_ in (ga, gd, gr, gw):
            g_.zero_()
        return gp, ga, gd, gr, (gw.sum().reshape(ema_w.shape) if shared else gw.reshape(ema_w.shape))
    with torch.cuda.device(dev):
        ws = _stage_ws(STAGE_PCEN, B, TP, F, 1, 1, dev)
        check(lib.leaf_pcen_backward_f32(_ptr(p), _ptr(g), B, F, TP, _ptr(alpha), _ptr(delta), _ptr(root), _ptr(w), float(floor),
                                         _ptr(gp), _ptr(ga), _ptr(gd), _ptr(gr), _ptr(gw), _ptr(ws), ws.numel(),
                                         stream_ptr(dev)), "leaf_pcen_backward_f32")
    return gp, ga, gd, gr, (gw.sum().reshape(ema_w.shape) if shared else gw.reshape(ema_w.shape))


def peak_normalize(x: torch.Tensor, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """Clips (rows of a (B,T) or (B,1,T) tensor) whose peak |x| exceeds 1 are divided by it; wraps leaf_peak_normalize_f32."""
    lib = load()
    require_hip(x, "peak_normalize")
    per_clip = 1
    for d in x.shape[1:]:
        per_clip *= int(d)
    x2 = _dev_f32(x.reshape(x.shape[0], per_clip), "x", x.device)        # (an explicit extent: -1 is ambiguous for B = 0)
    B, T = x2.shape
    if out is None:
        out = torch.empty_like(x2)
    elif out.dtype != torch.float32 or not out.is_contiguous() or out.numel() != x2.numel():
        raise RuntimeError("out must be a contiguous float32 tensor of the input's size")
    if B == 0:
        return out.reshape(x.shape)
    with torch.cuda.device(x.device):
        check(lib.leaf_peak_normalize_f32(_ptr(x2), B, T, _ptr(out), stream_ptr(x.device)), "leaf_peak_normalize_f32")
    return out.reshape(x.shape)


def prepare_tables(kernel: torch.Tensor, pool_w: torch.Tensor, K: int, hop: int) -> Optional[torch.Tensor]:
    """Parameter-derived tables of the overlap-save path (filter spectra + pooling rows) for frozen-parameter inference;
    ``None`` when that path does not cover the geometry.  Wraps leaf_fft_prepare_tables_f32."""
    lib = load()
    require_hip(kernel, "prepare_tables")
    dev = kernel.device
    kernel = _dev_f32(kernel, "kernel", dev)
    pw = _dev_f32(pool_w.reshape(-1), "pool_w", dev)
    F = kernel.shape[0]
    nbytes = lib.leaf_fft_tables_bytes(F, K, hop)
    if nbytes == 0:
        return None
    tables = torch.empty(nbytes, dtype=torch.uint8, device=dev)
    with torch.cuda.device(dev):
        check(lib.leaf_fft_prepare_tables_f32(_ptr(kernel), _ptr(pw), F, K, hop, _ptr(tables), nbytes, stream_ptr(dev)),
              "leaf_fft_prepare_tables_f32")
    return tables


def leaf_forward_prepared(x: torch.Tensor, tables: torch.Tensor, pool_b, alpha, delta, root, ema_w, F: int, K: int, hop: int,
                          pcen: bool = True, log1p: bool = False, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """Forward with tables from ``prepare_tables`` (same outputs as ``leaf_forward``, without the table kernel)."""
    lib = load()
    require_hip(x, "leaf_forward_prepared")
    if x.dim() == 3:
        if x.shape[1] != 1:
            raise RuntimeError(f"expected input of shape (B,1,T), got {tuple(x.shape)}")
        x2 = x[:, 0, :]
    elif x.dim() == 2:
        x2 = x
    else:
        raise RuntimeError(f"expected input of shape (B,1,T), got {tuple(x.shape)}")
    dev = x.device
    io_bf16 = x2.dtype == torch.bfloat16
    x2 = x2.detach().contiguous() if io_bf16 else _dev_f32(x2, "x", dev)
    B, T = x2.shape
    pool_b = _dev_f32(pool_b, "pool_b", dev)
    flags = FLAG_IO_BF16 if io_bf16 else 0
    if pcen:
        flags |= FLAG_PCEN
        alpha, delta, root, ema_w = (_dev_f32(t, "pcen param", dev) for t in (alpha, delta, root, ema_w))
    else:
        alpha = delta = root = ema_w = None
        if log1p:
            flags |= FLAG_LOG1P
    TP = lib.leaf_num_frames(T, K, hop)
    if out is None:
        out = torch.empty((B, F, TP), dtype=torch.bfloat16 if io_bf16 else torch.float32, device=dev)
    with torch.cuda.device(dev):
        ws = workspace(lib.leaf_workspace_bytes(B, T, F, K, hop, ALGO_FFT), dev)
        check(lib.leaf_forward_prepared_f32(_ptr(x2), B, T, _ptr(tables), tables.numel(), _ptr(pool_b), _ptr(alpha),
                                            _ptr(delta), _ptr(root), _ptr(ema_w), F, K, hop, flags, _ptr(out), _ptr(ws),
                                            ws.numel(), stream_ptr(dev)), "leaf_forward_prepared_f32")
    return out
