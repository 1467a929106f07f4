// leaf_common.hpp -- tuning knobs, vector types, Gabor tap / pooling-window formulas and their table kernels
// Part of the single translation unit leaf_kernels.hip (gfx950 only); see that file's header comment.
#pragma once
#include <hip/hip_runtime.h>
#include "leaf_fastmath.hpp"
#include <stdint.h>
#include <math.h>
#include <algorithm>
#include "leaf_hip.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x4u __attribute__((ext_vector_type(4), aligned(4)));   // 16-byte load at 4-byte alignment

constexpr float kPooledFloor = 1e-5f;    // frontend.py:84
// frontend.py:84 is torch.maximum(pooled, 1e-5): NaN propagates (fmaxf would return the floor instead).
__device__ __forceinline__ float pooled_floor(float v) { return v < kPooledFloor ? kPooledFloor : v; }
// Compile-time tuning knobs (tools/ablate.py builds variants of this file with -D...; the product uses the defaults)
#ifndef LEAF_WAVES_PER_WG
#define LEAF_WAVES_PER_WG 8
#endif
#ifndef LEAF_ABLATE
#define LEAF_ABLATE 0                    // bit0 skip epilogue, bit1 skip window staging, bit2 skip partial stores
#endif
#ifndef LEAF_KLOOP_SINGLE_BUFFER_RT
#define LEAF_KLOOP_SINGLE_BUFFER_RT 4    // register tiles with >= this many filter tiles use a single-buffered k-loop
#endif
#ifndef LEAF_DMA_PREFETCH
#define LEAF_DMA_PREFETCH 1              // next task's waveform window via global_load_lds under the epilogue
#endif
#ifndef LEAF_TRACE
#define LEAF_TRACE 0                     // tools/trace.py: per-phase s_memtime stamps of block 0 into the workspace tail
#endif
constexpr int kAblate = LEAF_ABLATE;
constexpr int kWavesPerWG = LEAF_WAVES_PER_WG;   // 8 -> 512 threads: 2 waves per SIMD
constexpr int kUB = 5;                   // 16-sample n-blocks per unit (register tile = RT x kUB MFMA tiles x2)
constexpr int kMaxLds = 160 * 1024;

struct GaborBounds { float sigma_lo, sigma_hi; };

// convolution.py:15-22 -- bounds are built from float32 tensors in the reference.
inline GaborBounds gabor_bounds(int K) {
    const float root = sqrtf(2.0f * logf(2.0f));
    GaborBounds b;
    b.sigma_lo = 4.0f * root / (float)M_PI;
    b.sigma_hi = (float)K * root / (float)M_PI;
    return b;
}

// impulse_responses.py:5-16 -- one complex Gabor tap at integer time t, from the UNclamped parameter.
// Same fp32 operation order as the reference: phase = fl(mu*t); env = exp(fl(1/(2 s^2)) * fl(-t^2)).
__device__ __forceinline__ void gabor_tap(float mu_raw, float sg_raw, GaborBounds bd, float t, float& re, float& im) {
    const float mu = fminf(fmaxf(mu_raw, 0.0f), 3.14159274101257324f);
    const float sg = fminf(fmaxf(sg_raw, bd.sigma_lo), bd.sigma_hi);
    const float norm = 1.0f / (2.50662827463100024f * sg);           // 1/(sqrt(2 pi) sigma)
    const float a = 1.0f / (2.0f * (sg * sg));
    const float env = expf(a * (-(t * t)));
    float s, c;
    sincosf(mu * t, &s, &c);
    re = (norm * c) * env;
    im = (norm * s) * env;
}

// ---------------------------------------------------------------------------------------------
// tap tables
// ---------------------------------------------------------------------------------------------

// Direct table, the layout convolution.py:88-90 hands to conv1d: taps[2f][j] = Re, taps[2f+1][j] = Im,
// t_j = j - K/2.
#ifndef LEAF_INST_TU               // non-template kernel: compiled once, in leaf_kernels.hip
__global__ void taps_direct_kernel(const float* __restrict__ kernel, int F, int K, GaborBounds bd,
                                   float* __restrict__ taps) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= F * K) return;
    const int f = idx / K, j = idx - f * K;
    float re, im;
    gabor_tap(kernel[2 * f], kernel[2 * f + 1], bd, (float)(j - K / 2), re, im);
    taps[(size_t)(2 * f) * K + j] = re;
    taps[(size_t)(2 * f + 1) * K + j] = im;
}
#endif

// impulse_responses.py:74-80
__device__ __forceinline__ float pool_sigma(float w_raw, int K) { return fminf(fmaxf(w_raw, 2.0f / (float)K), 0.5f); }

#ifndef LEAF_INST_TU               // non-template kernel: compiled once, in leaf_kernels.hip
__global__ void lowpass_window_kernel(const float* __restrict__ pool_w, int F, int K, float* __restrict__ g) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= F * K) return;
    const int f = idx / K, j = idx - f * K;
    const float half = 0.5f * (float)(K - 1);
    const float q = ((float)j - half) / (pool_sigma(pool_w[f], K) * half);
    g[idx] = expf(-0.5f * (q * q));
}
#endif

}  // namespace
