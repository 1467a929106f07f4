// leaf_stage_backward.hpp -- backward kernels of the stand-alone stage entry points (one reference module each)
// Part of the single translation unit leaf_kernels.hip (gfx950 only); see that file's header comment.
//
// The reference's sub-modules are ordinary differentiable nn.Modules (convolution.py:71-99, frontend.py:15-19,
// pooling.py:31-42, postprocessing.py:13-28,62-69).  Leaf.forward has its own fused backward (leaf_backward_f32); these
// kernels serve a sub-module that is called on its own under autograd.  One lane per output, every intermediate
// materialised -- the same correctness-first style as leaf_staged.hpp; clamp sub-gradients as torch.clamp gives them.
#pragma once
#include "leaf_common.hpp"

namespace {

// frontend.py:15-19 backward: dy[b,2f,n] = 2 y_re de, dy[b,2f+1,n] = 2 y_im de
__global__ void sqmod_bwd_kernel(const float* __restrict__ y, const float* __restrict__ ge, size_t BF, int T,
                                 float* __restrict__ gy) {
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= BF * (size_t)T) return;
    const size_t bf = idx / T;
    const int n = (int)(idx - bf * T);
    const float g2 = 2.0f * ge[idx];
    gy[(2 * bf) * T + n] = g2 * y[(2 * bf) * T + n];
    gy[(2 * bf + 1) * T + n] = g2 * y[(2 * bf + 1) * T + n];
}

// pooling.py:41 transposed w.r.t. the input: de[b,f,n] = sum_m g[f][n + padL - m hop] * gp[b,f,m]
__global__ void pool_bwd_de_kernel(const float* __restrict__ g, const float* __restrict__ gpooled, int F, int T, int TP,
                                   int K, int hop, int padL, float* __restrict__ de) {
    const int n = blockIdx.x * blockDim.x + threadIdx.x;
    const int f = blockIdx.y, b = blockIdx.z;
    if (n >= T) return;
    const float* w = g + (size_t)f * K;
    const float* gp = gpooled + ((size_t)b * F + f) * TP;
    const int np = n + padL;
    const int m_hi = min(TP - 1, np / hop);
    const int m_lo = max(0, (np - K + hop) / hop);
    float acc = 0.0f;
    for (int m = m_lo; m <= m_hi; ++m) {
        const int j = np - m * hop;
        if (j >= 0 && j < K) acc = fmaf(w[j], gp[m], acc);
    }
    de[((size_t)b * F + f) * T + n] = acc;
}

// postprocessing.py:13-28 backward, one lane per (b,f) row: M_m = w p_m + (1-w) M_{m-1}, M_{-1} = p_0.
// gM_m = g_m + (1-w) gM_{m+1};  dp_m = w gM_m (+ (1-w) gM_0 at m = 0);  dw = sum_m gM_m (p_m - M_{m-1}).
// scratch: [BF][TP] floats (the recomputed EMA).  rowsum[row] = this row's dw (clamp sub-gradient applied).
__global__ void ema_bwd_rows_kernel(const float* __restrict__ p, const float* __restrict__ gema, int BF, int F, int TP,
                                    const float* __restrict__ ema_w, float* __restrict__ scratch, float* __restrict__ gp,
                                    float* __restrict__ rowsum) {
    const int row = blockIdx.x * blockDim.x + threadIdx.x;
    if (row >= BF) return;
    const int f = row % F;
    const float wr = ema_w[f];
    const float w = fminf(fmaxf(wr, 0.0f), 1.0f), omw = 1.0f - w;
    const float* pr = p + (size_t)row * TP;
    const float* g = gema + (size_t)row * TP;
    float* M = scratch + (size_t)row * TP;
    float* o = gp + (size_t)row * TP;
    float state = pr[0];
    for (int m = 0; m < TP; ++m) {
        state = w * pr[m] + omw * state;
        M[m] = state;
    }
    float gM_next = 0.0f, s_w = 0.0f;
    for (int m = TP - 1; m >= 0; --m) {
        const float gM = g[m] + omw * gM_next;
        const float Mprev = m > 0 ? M[m - 1] : pr[0];
        s_w += gM * (pr[m] - Mprev);
        o[m] = w * gM + (m == 0 ? omw * gM : 0.0f);
        gM_next = gM;
    }
    rowsum[row] = (wr >= 0.0f && wr <= 1.0f) ? s_w : 0.0f;
}

// out[q][f] = sum_b rows[(b F + f) * stride + q]   (per-filter sum over the batch of per-row partials)
__global__ void rows_to_filter_sum_kernel(const float* __restrict__ rows, int B, int F, int stride, int nq,
                                          float* __restrict__ o0, float* __restrict__ o1, float* __restrict__ o2,
                                          float* __restrict__ o3) {
    __shared__ float red[256];
    const int f = blockIdx.x, tid = threadIdx.x;
    float* outs[4] = {o0, o1, o2, o3};
    for (int q = 0; q < nq; ++q) {
        float acc = 0.0f;
        for (int b = tid; b < B; b += 256) acc += rows[((size_t)b * F + f) * stride + q];
        red[tid] = acc;
        __syncthreads();
        for (int s = 128; s > 0; s >>= 1) {
            if (tid < s) red[tid] += red[tid + s];
            __syncthreads();
        }
        if (tid == 0 && outs[q]) outs[q][f] = red[0];
        __syncthreads();
    }
}

}  // namespace
