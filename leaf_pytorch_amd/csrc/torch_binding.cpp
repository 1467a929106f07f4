// torch_binding.cpp -- the "thin torch cpp_extension" over the C ABI (include/leaf_hip.h): registers the fused forward,
// the training forward and the backward as dispatcher ops in the `leaf_amd` namespace, so that
//   * a model containing leaf_pytorch_amd.Leaf traces under torch.compile / torch.export without a graph break
//     (fake kernels + the autograd formula are attached from Python, leaf_pytorch_amd/_ops.py), and
//   * an eager call costs one dispatcher hop instead of ~20 ctypes argument conversions.
// No arithmetic lives here: tensors are checked, outputs and the scratch workspace come from the caching allocator, the
// current HIP stream is handed through, and the matching C-ABI entry point does the work (no fallback of any kind).
// Plain C++ (compiled with g++ against the torch headers); reference counterpart: leaf_pytorch/frontend.py:78-89 and
// what autograd derives for it.
#include <algorithm>

#include <torch/library.h>
#include <ATen/ATen.h>
#include <ATen/hip/impl/HIPGuardImplMasqueradingAsCUDA.h>      // PyTorch-ROCm presents HIP devices under the "cuda" device type
#include <ATen/hip/impl/HIPStreamMasqueradingAsCUDA.h>

#include "leaf_hip.h"

namespace {

using at::Tensor;
using OptTensor = c10::optional<Tensor>;

const float* fptr(const Tensor& t) { return t.data_ptr<float>(); }
const float* fptr(const OptTensor& t) { return t.has_value() && t->defined() ? t->data_ptr<float>() : nullptr; }

Tensor dev_f32(const Tensor& t, const char* name, const c10::Device& dev) {
    TORCH_CHECK(t.device() == dev, name, " is on ", t.device(), ", expected ", dev);
    TORCH_CHECK(t.scalar_type() == at::kFloat, name, " must be float32, got ", t.scalar_type());
    return t.contiguous();
}
OptTensor dev_f32(const OptTensor& t, const char* name, const c10::Device& dev) {
    if (!t.has_value() || !t->defined()) return c10::nullopt;
    return dev_f32(*t, name, dev);
}

void check_status(int rc, const char* what) {
    TORCH_CHECK(rc == LEAF_OK, what, " failed: ", leaf_status_string(rc), " (status ", rc, ")");
}

// (B,1,T) or (B,T) -> contiguous (B,T) view of the waveform
Tensor waveform_2d(const Tensor& x) {
    TORCH_CHECK(x.is_cuda(), "leaf_amd: input is on '", x.device(),
                "'. leaf_pytorch_amd runs only on an AMD GPU through its HIP kernels; there is no CPU path in the product");
    TORCH_CHECK(x.dim() == 2 || (x.dim() == 3 && x.size(1) == 1), "expected input of shape (B,1,T), got ", x.sizes());
    return (x.dim() == 3 ? x.select(1, 0) : x).contiguous();
}

// Slices of whole clips for batches beyond one C-ABI call (B * T < 2^31 per call): as few calls as possible, balanced.
struct BatchSlices { int64_t per_call, calls; };
BatchSlices batch_slices(int64_t B, int64_t T) {
    const int64_t most = std::max<int64_t>(1, ((int64_t(1) << 31) - 1) / std::max<int64_t>(T, 1));
    const int64_t calls = std::max<int64_t>(1, (B + most - 1) / most);
    return {(B + calls - 1) / calls, calls};
}

struct Params {
    Tensor kernel, pool_w, pool_b;
    OptTensor alpha, delta, root, ema_w;
    bool pcen;
};
Params gather(const Tensor& kernel, const Tensor& pool_w, const Tensor& pool_b, const OptTensor& alpha, const OptTensor& delta,
              const OptTensor& root, const OptTensor& ema_w, const c10::Device& dev) {
    Params p;
    p.kernel = dev_f32(kernel, "kernel", dev);
    p.pool_w = dev_f32(pool_w.reshape({-1}), "pool_w", dev);
    p.pool_b = dev_f32(pool_b, "pool_b", dev);
    p.pcen = alpha.has_value() && alpha->defined();
    if (p.pcen) {
        p.alpha = dev_f32(alpha, "alpha", dev); p.delta = dev_f32(delta, "delta", dev);
        p.root = dev_f32(root, "root", dev); p.ema_w = dev_f32(ema_w, "ema_w", dev);
        TORCH_CHECK(p.delta && p.root && p.ema_w, "PCEN needs alpha, delta, root and ema_w");
    }
    return p;
}

Tensor forward_impl(const Tensor& x, const Params& p, int64_t K, int64_t hop, bool log1p, int64_t algo, Tensor* raw) {
    Tensor x2 = waveform_2d(x);
    const bool io_bf16 = x2.scalar_type() == at::kBFloat16;
    TORCH_CHECK(io_bf16 || x2.scalar_type() == at::kFloat, "x must be float32 (or bfloat16 for the bf16-I/O extension), got ",
                x2.scalar_type());
    TORCH_CHECK(x2.size(1) < (int64_t(1) << 31), "a clip of ", x2.size(1), " samples is beyond the C ABI's 32-bit sample index");
    const int64_t B = x2.size(0);
    const int T = (int)x2.size(1), F = (int)p.kernel.size(0);
    const int TP = leaf_num_frames(T, (int)K, (int)hop);
    TORCH_CHECK(TP >= 1 && F >= 1, "bad shape B=", B, " T=", T, " F=", F, " K=", K, " hop=", hop);
    if (B == 0) {
        // the empty batch: the reference returns (0, F, T') (frontend.py:78-89 -> convolution.py:97); nothing is launched
        if (raw) *raw = at::empty({0, F, TP}, x2.options().dtype(at::kFloat));
        return at::empty({0, F, TP}, x2.options());
    }
    int flags = (io_bf16 ? LEAF_FLAG_IO_BF16 : 0) | (p.pcen ? LEAF_FLAG_PCEN : (log1p ? LEAF_FLAG_LOG1P : 0));
    // call options travelling in the upper bits of the op's `algo` argument (the schema stays as it is): bit 24 = the
    // PeakNormalization prologue folded into the forward (LEAF_FLAG_PEAKNORM; inference only)
    constexpr int64_t kOptPeakNorm = int64_t(1) << 24;
    if (algo & kOptPeakNorm) {
        TORCH_CHECK(!raw, "the fused PeakNormalization prologue is forward-only");
        flags |= LEAF_FLAG_PEAKNORM;
        algo &= ~kOptPeakNorm;
    }
    c10::hip::HIPGuardMasqueradingAsCUDA guard(x2.device());
    auto stream = c10::hip::getCurrentHIPStreamMasqueradingAsCUDA(x2.device().index());
    Tensor out = at::empty({B, F, TP}, x2.options());
    if (raw) *raw = at::empty({B, F, TP}, x2.options().dtype(at::kFloat));
    // The C ABI indexes the samples of ONE call with 32 bits and refuses B * T >= 2^31 (LEAF_ERR_BAD_SHAPE); the reference's
    // conv1d takes any batch (frontend.py:78-89).  Clips are independent, so a larger batch goes through in balanced slices of
    // whole clips, one C-ABI call each, into the one preallocated output: the bits of a clip do not depend on its slice (within
    // one kernel family; the slices are far beyond every AUTO threshold).  One workspace, sized for the largest slice, serves
    // the stream-ordered calls in turn.
    const BatchSlices sl = batch_slices(B, T);
    const int last = (int)(B - (sl.calls - 1) * sl.per_call);
    Tensor ws = at::empty({(int64_t)std::max<size_t>({leaf_workspace_bytes((int)sl.per_call, T, F, (int)K, (int)hop, (int)algo),
                                                      leaf_workspace_bytes(last, T, F, (int)K, (int)hop, (int)algo), size_t(4)})},
                          x2.options().dtype(at::kByte));
    const size_t io = io_bf16 ? 2 : 4;
    for (int64_t b0 = 0; b0 < B; b0 += sl.per_call) {
        const int nb = (int)std::min<int64_t>(sl.per_call, B - b0);
        const float* xin = reinterpret_cast<const float*>(static_cast<const char*>(x2.data_ptr()) + (size_t)b0 * T * io);
        float* o = reinterpret_cast<float*>(static_cast<char*>(out.data_ptr()) + (size_t)b0 * F * TP * io);
        int rc;
        if (raw) {
            rc = leaf_forward_save_f32(xin, nb, T, fptr(p.kernel), fptr(p.pool_w), fptr(p.pool_b), fptr(p.alpha), fptr(p.delta),
                                       fptr(p.root), fptr(p.ema_w), F, (int)K, (int)hop, flags, (int)algo, o,
                                       raw->data_ptr<float>() + (size_t)b0 * F * TP, ws.data_ptr(), (size_t)ws.numel(), stream.stream());
            check_status(rc, "leaf_forward_save_f32");
        } else {
            rc = leaf_forward_f32(xin, nb, T, fptr(p.kernel), fptr(p.pool_w), fptr(p.pool_b), fptr(p.alpha), fptr(p.delta),
                                  fptr(p.root), fptr(p.ema_w), F, (int)K, (int)hop, flags, (int)algo, o, ws.data_ptr(),
                                  (size_t)ws.numel(), stream.stream());
            check_status(rc, "leaf_forward_f32");
        }
    }
    return out;
}

// leaf_amd::forward -- frontend.py:78-89 (inference / no-grad)
Tensor op_forward(const Tensor& x, const Tensor& kernel, const Tensor& pool_w, const Tensor& pool_b, const OptTensor& alpha,
                  const OptTensor& delta, const OptTensor& root, const OptTensor& ema_w, int64_t K, int64_t hop, bool log1p,
                  int64_t algo) {
    const Params p = gather(kernel, pool_w, pool_b, alpha, delta, root, ema_w, x.device());
    return forward_impl(x, p, K, hop, log1p, algo, nullptr);
}

// leaf_amd::forward_train -- the same, additionally returning the pre-floor pooled tensor the backward consumes
std::tuple<Tensor, Tensor> op_forward_train(const Tensor& x, const Tensor& kernel, const Tensor& pool_w, const Tensor& pool_b,
                                            const OptTensor& alpha, const OptTensor& delta, const OptTensor& root,
                                            const OptTensor& ema_w, int64_t K, int64_t hop, int64_t algo) {
    const Params p = gather(kernel, pool_w, pool_b, alpha, delta, root, ema_w, x.device());
    Tensor raw;
    Tensor out = forward_impl(x, p, K, hop, false, algo, &raw);
    return {out, raw};
}

// leaf_amd::backward -- what autograd derives for frontend.py:78-89: (g_kernel, g_pool_w, g_pool_b, g_alpha, g_delta, g_root,
// g_ema_w, g_x); the PCEN entries are empty tensors without PCEN, g_x is empty unless need_dx.
std::vector<Tensor> op_backward(const Tensor& x, const Tensor& kernel, const Tensor& pool_w, const Tensor& pool_b,
                                const OptTensor& alpha, const OptTensor& delta, const OptTensor& root, const OptTensor& ema_w,
                                int64_t K, int64_t hop, const Tensor& grad_out, const OptTensor& pooled_raw, bool need_dx,
                                int64_t flags) {
    Tensor x2 = waveform_2d(x);
    TORCH_CHECK(x2.scalar_type() == at::kFloat, "the backward is float32 only, got ", x2.scalar_type());
    const Params p = gather(kernel, pool_w, pool_b, alpha, delta, root, ema_w, x.device());
    TORCH_CHECK(x2.size(1) < (int64_t(1) << 31), "a clip of ", x2.size(1), " samples is beyond the C ABI's 32-bit sample index");
    const int64_t B = x2.size(0);
    const int T = (int)x2.size(1), F = (int)p.kernel.size(0);
    const int TP = leaf_num_frames(T, (int)K, (int)hop);
    Tensor go = dev_f32(grad_out, "grad_out", x2.device());
    TORCH_CHECK(go.dim() == 3 && go.size(0) == B && go.size(1) == F && go.size(2) == TP, "grad_out has shape ", go.sizes(),
                ", expected (", B, ",", F, ",", TP, ")");
    OptTensor raw = dev_f32(pooled_raw, "pooled_raw", x2.device());
    c10::hip::HIPGuardMasqueradingAsCUDA guard(x2.device());
    auto stream = c10::hip::getCurrentHIPStreamMasqueradingAsCUDA(x2.device().index());
    auto opt = x2.options();
    Tensor gk = at::empty_like(p.kernel), gpw = at::empty_like(p.pool_w), gpb = at::empty_like(p.pool_b);
    Tensor ga = at::empty({p.pcen ? F : 0}, opt), gd = at::empty({p.pcen ? F : 0}, opt), gr = at::empty({p.pcen ? F : 0}, opt),
           gw = at::empty({p.pcen ? F : 0}, opt);
    Tensor gx = need_dx ? at::empty_like(x2) : at::empty({0}, opt);
    if (B == 0) {                                             // the sum over no clips (C ABI: zero-fills, launches nothing else)
        for (Tensor* g : {&gk, &gpw, &gpb, &ga, &gd, &gr, &gw}) g->zero_();
        return {gk, gpw.reshape(pool_w.sizes()), gpb, ga, gd, gr, gw, need_dx ? gx.reshape(x.sizes()) : gx};
    }
    const int fl = (int)flags | (p.pcen ? LEAF_FLAG_PCEN : 0);
    // B * T >= 2^31: slices of whole clips as in the forward; the parameter gradients of the slices are added in slice order
    // (a fixed order: the step stays bit-reproducible), dL/dx is written slice by slice
    const BatchSlices sl = batch_slices(B, T);
    const int last = (int)(B - (sl.calls - 1) * sl.per_call);
    Tensor ws = at::empty({(int64_t)std::max<size_t>({leaf_backward_workspace_bytes((int)sl.per_call, T, F, (int)K, (int)hop, fl, need_dx ? 1 : 0),
                                                      leaf_backward_workspace_bytes(last, T, F, (int)K, (int)hop, fl, need_dx ? 1 : 0), size_t(4)})},
                          opt.dtype(at::kByte));
    Tensor tk, tpw, tpb, ta, td, tr, tw;
    if (sl.calls > 1) {
        tk = at::empty_like(gk); tpw = at::empty_like(gpw); tpb = at::empty_like(gpb);
        ta = at::empty_like(ga); td = at::empty_like(gd); tr = at::empty_like(gr); tw = at::empty_like(gw);
    }
    for (int64_t b0 = 0; b0 < B; b0 += sl.per_call) {
        const int nb = (int)std::min<int64_t>(sl.per_call, B - b0);
        const bool first = b0 == 0;
        Tensor &k_ = first ? gk : tk, &pw_ = first ? gpw : tpw, &pb_ = first ? gpb : tpb, &a_ = first ? ga : ta, &d_ = first ? gd : td,
               &r_ = first ? gr : tr, &w_ = first ? gw : tw;
        const int rc = leaf_backward_f32(fptr(x2) + (size_t)b0 * T, nb, T, fptr(p.kernel), fptr(p.pool_w), fptr(p.pool_b), fptr(p.alpha),
                                         fptr(p.delta), fptr(p.root), fptr(p.ema_w), F, (int)K, (int)hop, fl,
                                         fptr(go) + (size_t)b0 * F * TP, raw ? fptr(raw) + (size_t)b0 * F * TP : nullptr,
                                         k_.data_ptr<float>(), pw_.data_ptr<float>(), pb_.data_ptr<float>(),
                                         p.pcen ? a_.data_ptr<float>() : nullptr, p.pcen ? d_.data_ptr<float>() : nullptr,
                                         p.pcen ? r_.data_ptr<float>() : nullptr, p.pcen ? w_.data_ptr<float>() : nullptr,
                                         need_dx ? gx.data_ptr<float>() + (size_t)b0 * T : nullptr, ws.data_ptr(), (size_t)ws.numel(),
                                         stream.stream());
        check_status(rc, "leaf_backward_f32");
        if (!first) {
            gk.add_(tk); gpw.add_(tpw); gpb.add_(tpb);
            if (p.pcen) { ga.add_(ta); gd.add_(td); gr.add_(tr); gw.add_(tw); }
        }
    }
    return {gk, gpw.reshape(pool_w.sizes()), gpb, ga, gd, gr, gw, need_dx ? gx.reshape(x.sizes()) : gx};
}

}  // namespace

TORCH_LIBRARY(leaf_amd, m) {
    m.def("forward(Tensor x, Tensor kernel, Tensor pool_w, Tensor pool_b, Tensor? alpha, Tensor? delta, Tensor? root, "
          "Tensor? ema_w, int K, int hop, bool log1p, int algo) -> Tensor");
    m.def("forward_train(Tensor x, Tensor kernel, Tensor pool_w, Tensor pool_b, Tensor? alpha, Tensor? delta, Tensor? root, "
          "Tensor? ema_w, int K, int hop, int algo) -> (Tensor, Tensor)");
    m.def("backward(Tensor x, Tensor kernel, Tensor pool_w, Tensor pool_b, Tensor? alpha, Tensor? delta, Tensor? root, "
          "Tensor? ema_w, int K, int hop, Tensor grad_out, Tensor? pooled_raw, bool need_dx, int flags) -> Tensor[]");
}

// HIP tensors dispatch under the CUDA key in PyTorch-ROCm
TORCH_LIBRARY_IMPL(leaf_amd, CUDA, m) {
    m.impl("forward", &op_forward);
    m.impl("forward_train", &op_forward_train);
    m.impl("backward", &op_backward);
}
