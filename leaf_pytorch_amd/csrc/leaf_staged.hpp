// leaf_staged.hpp -- staged (one kernel per reference module) forward kernels
// Part of the single translation unit leaf_kernels.hip (gfx950 only); see that file's header comment.
#pragma once
#include "leaf_common.hpp"

namespace {

// ---------------------------------------------------------------------------------------------
// staged (unfused) kernels: one per reference module.  Correctness-first; used by the sub-modules
// when called on their own, as the on-device cross-check of the fused kernel, and as the fallback
// for geometries the fused kernel does not cover.
// ---------------------------------------------------------------------------------------------

// convolution.py:91-97 -- y[b][c][n] = sum_j taps[c][j] * xz[b][n + j - padL]
__global__ void conv_staged_kernel(const float* __restrict__ x, const float* __restrict__ taps, int B, int T,
                                   int C, int K, int padL, float* __restrict__ y) {
    (void)B;
    const int n = blockIdx.x * blockDim.x + threadIdx.x;
    const int c = blockIdx.y, b = blockIdx.z;
    if (n >= T) return;
    const float* xb = x + (size_t)b * T;
    const float* w = taps + (size_t)c * K;
    float acc = 0.0f;
    const int j0 = max(0, padL - n), j1 = min(K, T + padL - n);
    for (int j = j0; j < j1; ++j) acc = fmaf(w[j], xb[n + j - padL], acc);
    y[((size_t)b * C + c) * T + n] = acc;
}

// frontend.py:15-19
__global__ void sqmod_kernel(const float* __restrict__ y, size_t BF, int T, float* __restrict__ e) {
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= BF * (size_t)T) return;
    const size_t bf = idx / T;
    const int n = (int)(idx - bf * T);
    const float re = y[(2 * bf) * T + n], im = y[(2 * bf + 1) * T + n];
    e[idx] = re * re + im * im;
}

// pooling.py:41 -- p[b][f][m] = bias_f + sum_j g[f][j] * ez[b][f][m*hop + j - padL]
__global__ void pool_staged_kernel(const float* __restrict__ e, const float* __restrict__ g,
                                   const float* __restrict__ bias, int F, int T, int TP, int K, int hop, int padL,
                                   float* __restrict__ pooled) {
    const int m = blockIdx.x * blockDim.x + threadIdx.x;
    const int f = blockIdx.y, b = blockIdx.z;
    if (m >= TP) return;
    const float* eb = e + ((size_t)b * F + f) * T;
    const float* w = g + (size_t)f * K;
    const int base = m * hop - padL;
    const int j0 = max(0, -base), j1 = min(K, T - base);
    float acc = 0.0f;
    for (int j = j0; j < j1; ++j) acc = fmaf(w[j], eb[base + j], acc);
    pooled[((size_t)b * F + f) * TP + m] = acc + (bias ? bias[f] : 0.0f);
}

// postprocessing.py:13-28 + 62-69 on a (B,F,T') tensor; one lane per (b,f) row.
// mode: 0 = EMA only, 1 = PCEN
__global__ void pcen_rows_kernel(const float* __restrict__ p, int BF, int F, int TP, const float* __restrict__ alpha,
                                 const float* __restrict__ delta, const float* __restrict__ root,
                                 const float* __restrict__ ema_w, float floor_, int mode, float* __restrict__ out) {
    const int row = blockIdx.x * blockDim.x + threadIdx.x;
    if (row >= BF) return;
    const int f = row % F;
    const float w = fminf(fmaxf(ema_w[f], 0.0f), 1.0f);
    const float omw = 1.0f - w;
    float a = 0.f, d = 0.f, inv_r = 0.f, d_r = 0.f;
    if (mode == 1) {
        a = fminf(alpha[f], 1.0f);
        inv_r = 1.0f / fmaxf(root[f], 1.0f);
        d = delta[f];
        d_r = powf(d, inv_r);
    }
    const float* pr = p + (size_t)row * TP;
    float* o = out + (size_t)row * TP;
    float state = pr[0];
    for (int m = 0; m < TP; ++m) {
        const float v = pr[m];
        state = w * v + omw * state;
        o[m] = (mode == 1) ? powf(v / powf(floor_ + state, a) + d, inv_r) - d_r : state;
    }
}

// Upstream transform of every reference data pipeline (utilities/data/raw_transforms.py:334-345: PeakNormalization with
// apply_to="only_too_loud_sounds"): a clip whose peak |x| exceeds 1 is divided by that peak, quieter clips pass unchanged.
// One workgroup per clip: wave-shuffle + LDS max reduction over coalesced float4 reads, then one scaled copy.
__global__ __launch_bounds__(1024) void peak_normalize_kernel(const float* __restrict__ x, int T, float* __restrict__ out) {
    __shared__ float red[16];
    const int b = blockIdx.x, tid = threadIdx.x;
    const float* xb = x + (size_t)b * T;
    float* ob = out + (size_t)b * T;
    float m = 0.0f;
    for (int i = tid; i < T; i += 1024) m = fmaxf(m, fabsf(xb[i]));
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) m = fmaxf(m, __shfl_xor(m, off));
    if ((tid & 63) == 0) red[tid >> 6] = m;
    __syncthreads();
    float peak = red[0];
#pragma unroll
    for (int w = 1; w < 16; ++w) peak = fmaxf(peak, red[w]);
    const float scale = peak > 1.0f ? 1.0f / peak : 1.0f;
    for (int i = tid; i < T; i += 1024) ob[i] = peak > 1.0f ? xb[i] * scale : xb[i];
}

// LEAF_FLAG_PEAKNORM: the per-clip scale of the transform above without the copy -- scale2[b] = s_b^2, s_b = 1 / peak when
// the clip's peak |x| exceeds 1, else 1 (what the overlap-save finalize multiplies the pooled energies by).  x fp32 or bf16.
#ifndef LEAF_INST_TU
__global__ __launch_bounds__(1024) void peak_scale2_kernel(const void* __restrict__ x_, int io_bf16, int T, float* __restrict__ scale2) {
    __shared__ float red[16];
    const int b = blockIdx.x, tid = threadIdx.x;
    float m = 0.0f;
    if (io_bf16) {
        const unsigned short* xb = static_cast<const unsigned short*>(x_) + (size_t)b * T;
        for (int i = tid; i < T; i += 1024) m = fmaxf(m, fabsf(__uint_as_float((unsigned)xb[i] << 16)));
    } else {
        const float* xb = static_cast<const float*>(x_) + (size_t)b * T;
        for (int i = tid; i < T; i += 1024) m = fmaxf(m, fabsf(xb[i]));
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) m = fmaxf(m, __shfl_xor(m, off));
    if ((tid & 63) == 0) red[tid >> 6] = m;
    __syncthreads();
    if (tid == 0) {
        float peak = red[0];
#pragma unroll
        for (int w = 1; w < 16; ++w) peak = fmaxf(peak, red[w]);
        const float s = peak > 1.0f ? 1.0f / peak : 1.0f;
        scale2[b] = s * s;
    }
}
#endif

}  // namespace
