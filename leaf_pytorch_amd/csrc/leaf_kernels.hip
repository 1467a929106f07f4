// leaf_kernels.hip -- MI355X (gfx950 / CDNA4) kernels for the LEAF frontend forward path, and the
// C ABI declared in include/leaf_hip.h.  Written for gfx950 only: 64-lane wavefronts, fp32 MFMA
// (v_mfma_f32_16x16x4_f32), 160 KiB LDS per CU.  No torch types anywhere in this file.
//
// Reference arithmetic being replaced (file:line under the reference repository):
//   leaf_pytorch/frontend.py:78-89        Leaf.forward
//   leaf_pytorch/convolution.py:15-22     GaborConstraint            -> constrain() below
//   leaf_pytorch/impulse_responses.py:5-16,66-71  Gabor taps         -> gabor_tap()
//   leaf_pytorch/convolution.py:71-99     GaborConv1d.forward        -> fused kernel / conv_staged
//   leaf_pytorch/frontend.py:15-19        SquaredModulus             -> fused kernel / sqmod
//   leaf_pytorch/impulse_responses.py:74-80 + pooling.py:31-42       -> fused kernel / pool_staged
//   leaf_pytorch/postprocessing.py:13-28,62-69  EMA + PCEN           -> finalize kernel
//
// Design (see DESIGN.md for the full derivation):
//   * The Gabor taps are Hermitian in t (Re even, Im odd), so with s_k[n] = x[n+k] + x[n-k] and
//     d_k[n] = x[n+k] - x[n-k] the complex filterbank is two real GEMMs with HALF the K extent:
//         Re y[n,f] = sum_k s_k[n] * hr_f[k],   Im y[n,f] = sum_k d_k[n] * hi_f[k],  k = 0..K/2
//     (even K: one extra row whose forward sample is masked).  Both GEMMs run on the fp32 MFMA
//     (exact fp32 fmaf chains at the fp32 vector rate, operands delivered from LDS).
//   * One wave owns one "hop-block" (hop consecutive output samples aligned with the pooling frame
//     grid) of one clip: it stages its own waveform window in LDS (no inter-wave sync in the main
//     loop), accumulates Re/Im tiles in registers, squares them, applies the Gaussian pooling
//     weights on the VALU (exp2 on the fly) and reduces to per-frame partial sums.  The 80x-inflated
//     (B,2F,T) tensor of the reference never exists.
//   * A second tiny kernel sums the <= NOFF partials per frame, adds the bias, floors at 1e-5 and
//     runs the PCEN recurrence.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <math.h>
#include <algorithm>

#include "leaf_hip.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x4u __attribute__((ext_vector_type(4), aligned(4)));   // 16-byte load at 4-byte alignment

constexpr float kPooledFloor = 1e-5f;    // frontend.py:84
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

// impulse_responses.py:74-80
__device__ __forceinline__ float pool_sigma(float w_raw, int K) { return fminf(fmaxf(w_raw, 2.0f / (float)K), 0.5f); }

__global__ void lowpass_window_kernel(const float* __restrict__ pool_w, int F, int K, float* __restrict__ g) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= F * K) return;
    const int f = idx / K, j = idx - f * K;
    const float half = 0.5f * (float)(K - 1);
    const float q = ((float)j - half) / (pool_sigma(pool_w[f], K) * half);
    g[idx] = expf(-0.5f * (q * q));
}

// ---------------------------------------------------------------------------------------------
// staged (unfused) kernels: one per reference module.  Correctness-first; used by the sub-modules
// when called on their own, as the on-device cross-check of the fused kernel, and as the fallback
// for geometries the fused kernel does not cover.
// ---------------------------------------------------------------------------------------------

// convolution.py:91-97 -- y[b][c][n] = sum_j taps[c][j] * xz[b][n + j - padL]
__global__ void conv_staged_kernel(const float* __restrict__ x, const float* __restrict__ taps, int B, int T,
                                   int C, int K, int padL, float* __restrict__ y) {
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

// ---------------------------------------------------------------------------------------------
// fused path: prep (filter ordering + half-support tap table), fused filterbank/pool kernel, finalize
// ---------------------------------------------------------------------------------------------

// Taps smaller than exp(-kTapCut^2/2) = 1.5e-8 of a filter's peak are not issued: the Gaussian envelope
// puts them below the fp32 rounding noise of the 400-term sums they would join.
constexpr float kTapCut = 6.0f;
constexpr int kMaxFP = 256;              // the fused path handles up to 256 (padded) filters

// One launch builds everything the fused kernel needs from the raw parameters:
//   perm[col]   filter index held by tap column col (columns are sorted by decreasing half-support so each
//               16-column MFMA tile groups filters of similar width); -1 for padding columns
//   col_of[f]   inverse map
//   tile_ks[t]  number of 4-row k-steps tile t needs = ceil((largest half-support in the tile + 1)/4)
//   W[kk][c]    c <  FP: Re tap of filter perm[c] at t=+kk;  c >= FP: Im tap of filter perm[c-FP]
//               (zero beyond that filter's own half-support, so a filter's result never depends on its tile
//               mates).  Row 0 carries hr[0]/2 because the kernel forms s_0 = x[n] + x[n].
//   G[c][j]     Gaussian pooling window of filter perm[c] (impulse_responses.py:74-80), j = 0..GJ-1, ZERO for
//               j >= K: the fused epilogue reads it with 16-byte loads and needs no window masks.
// Every block recomputes the (tiny) ordering in LDS; block 0 publishes it.
__global__ __launch_bounds__(256) void fused_prep_kernel(const float* __restrict__ kernel,
                                                         const float* __restrict__ pool_w, int F, int FP, int K, int R,
                                                         int GJ, GaborBounds bd, float* __restrict__ W,
                                                         float* __restrict__ G, float* __restrict__ Gs,
                                                         int* __restrict__ perm, int* __restrict__ col_of,
                                                         int* __restrict__ tile_ks) {
    __shared__ int s_sup[kMaxFP];        // half-support per filter slot (-1 = padding)
    __shared__ int s_perm[kMaxFP];
    const int tid = threadIdx.x;
    const int Hb = K / 2;
    for (int c = tid; c < FP; c += 256) {
        int sup = -1;
        if (c < F) {
            const float sg = fminf(fmaxf(kernel[2 * c + 1], bd.sigma_lo), bd.sigma_hi);
            sup = min(Hb, (int)ceilf(kTapCut * sg));
        }
        s_sup[c] = sup;
    }
    __syncthreads();
    for (int c = tid; c < FP; c += 256) {
        const int mine = s_sup[c];
        int rank = 0;
        for (int o = 0; o < FP; ++o) {
            const int other = s_sup[o];
            rank += (other > mine) || (other == mine && o < c);
        }
        s_perm[rank] = c;
    }
    __syncthreads();
    if (blockIdx.x == 0) {
        for (int c = tid; c < FP; c += 256) {
            const int f = s_perm[c];
            perm[c] = f < F ? f : -1;
            if (f < F) col_of[f] = c;
            if ((c & 15) == 0) tile_ks[c >> 4] = (s_sup[f] + 1 + 3) / 4;     // sorted: first column of a tile is its widest
        }
    }
    int idx = blockIdx.x * 256 + tid;
    const int ncol = 2 * FP;
    if (idx >= R * ncol) {
        idx -= R * ncol;
        if (idx < FP * GJ) {
            const int c = idx / GJ, j = idx - c * GJ;
            const int f = s_perm[c];
            float v = 0.0f, dv = 0.0f;
            if (f < F && j < K) {
                const float half = 0.5f * (float)(K - 1);
                const float sig = pool_sigma(pool_w[f], K);
                const float q = ((float)j - half) / (sig * half);
                v = expf(-0.5f * (q * q));
                dv = v * (q * q) / sig;                  // d g / d s = g (j-c)^2 / (c^2 s^3)
            }
            G[idx] = v;
            if (Gs) Gs[idx] = dv;                        // backward only
        }
        return;
    }
    const int kk = idx / ncol, col = idx - kk * ncol;
    const bool is_im = col >= FP;
    const int f = s_perm[is_im ? col - FP : col];
    float v = 0.0f;
    if (f < F && kk <= s_sup[f]) {
        float re, im;
        gabor_tap(kernel[2 * f], kernel[2 * f + 1], bd, (float)kk, re, im);
        v = is_im ? im : re;
        if (kk == 0) v *= 0.5f;
    }
    W[idx] = v;
}

struct FusedParams {
    const void* x;         // [B][T] fp32, or bf16 when io_bf16
    int io_bf16;
    const float* W;        // [R][2*FP] half-support tap table (columns in perm order)
    const float* G;        // [FP][GJ] pooling windows (columns in perm order), zero for j >= K
    const int* tile_ks;    // [FP/16]
    int GJ;                // row length of G: noff*hop + 16*kUB*NU rounded up to 4
    float* part;           // [B][TP][noff][FP] per-frame partial pooled sums (columns in perm order)
    int B, T, TP, F, FP, K, hop, padL;
    int KS;                // k-steps of 4 rows, R = 4*KS
    int Hf;                // largest kk whose forward sample x[n+kk] is a real tap: (K-1)/2
    int xshift;            // K/2 - padL: 0 for odd K, 1 for even K
    int NU;                // units of kUB n-blocks per hop-block
    int HP;                // halo (floats) on each side of a wave's staged window = 4*KS
    int XS;                // floats per wave window = 16*kUB*NU + 2*HP
    int q_lo, nq;          // hop-blocks q_lo .. q_lo+nq-1 cover the samples of one clip
    int noff;              // frames a hop-block contributes to: (K-1)/hop + 1
    int tile_base;         // first 16-filter tile of this launch
    int total_tasks;       // B * nq
    int desync_sleeps;     // s_sleep(127) repetitions the second wave of each SIMD waits once at start
    unsigned long long* trace;   // LEAF_TRACE builds only: [8 waves][64] cycle stamps of block 0
    // backward instantiation (BWD) only:
    const float* Gs;       // [FP][GJ] d g/d s tables (same layout as G)
    const float* gcols;    // [B][TP][FP] grad w.r.t. the pre-floor pooled value, columns in perm order
    float* dY;             // [B*T][2*FP] out: grad w.r.t. the filterbank output, time-major, columns as W
    float* dwpart;         // [gridDim.x*kWavesPerWG][FP] out: per-wave partial sums of d pool_w (pre clamp mask)
};


// k-steps [ks, ks_end) of one unit with the first NA (widest) tiles of the workgroup active.
// Operands of step ks+1 are fetched from LDS into a second register set while the MFMAs of step ks issue.
template <int RT, int NA, bool EVENK>
struct KStep {
    float af[kUB], ab[kUB], bre[NA], bim[NA];
    __device__ __forceinline__ void load(const float* xf, const float* xb_, const float* sW, int offE, int offO, int ks) {
        constexpr int NC = 32 * RT;
        const int kk0 = 4 * ks;
        const float* wrow = sW + (size_t)kk0 * NC;
#pragma unroll
        for (int t = 0; t < NA; ++t) {
            bre[t] = wrow[((t & 1) ? offO : offE) + 16 * t];
            bim[t] = wrow[(((RT + t) & 1) ? offO : offE) + 16 * (RT + t)];
        }
#pragma unroll
        for (int nb = 0; nb < kUB; ++nb) {
            af[nb] = xf[16 * nb + kk0];
            ab[nb] = xb_[16 * nb - kk0];
        }
    }
    // FIRST: this is k-step 0 of a unit -- the accumulators start from the MFMA's inline-constant zero C operand
    // instead of being cleared by 4*2*RT*kUB v_mov (VALU time is not hidden under fp32 MFMAs on gfx950).
    template <bool FIRST = false>
    __device__ __forceinline__ void mma(f32x4 (&acc_re)[RT][kUB], f32x4 (&acc_im)[RT][kUB], int g, int Hf, int ks) const {
        const f32x4 zero = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int nb = 0; nb < kUB; ++nb) {
            float fw = af[nb];
            if (EVENK) fw = (4 * ks + g) <= Hf ? fw : 0.0f;   // the lone tap t = -K/2 of an even window
            const float s = fw + ab[nb], d = fw - ab[nb];
#pragma unroll
            for (int t = 0; t < NA; ++t) {
                acc_re[t][nb] = __builtin_amdgcn_mfma_f32_16x16x4f32(s, bre[t], FIRST ? zero : acc_re[t][nb], 0, 0, 0);
                acc_im[t][nb] = __builtin_amdgcn_mfma_f32_16x16x4f32(d, bim[t], FIRST ? zero : acc_im[t][nb], 0, 0, 0);
            }
        }
    }
};

template <int RT, int NA, bool EVENK, bool FIRSTSEG = false>
__device__ __forceinline__ void fused_ksegment(f32x4 (&acc_re)[RT][kUB], f32x4 (&acc_im)[RT][kUB], const float* xf,
                                               const float* xb_, const float* sW, int offE, int offO, int g, int Hf,
                                               int& ks, int ks_end) {
    if (ks >= ks_end) {
        if constexpr (FIRSTSEG) {                        // degenerate: no k-steps at all -> accumulators are zero
#pragma unroll
            for (int t = 0; t < RT; ++t)
#pragma unroll
                for (int nb = 0; nb < kUB; ++nb) acc_re[t][nb] = acc_im[t][nb] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
        return;
    }
    if constexpr (LEAF_KLOOP_SINGLE_BUFFER_RT <= RT) {
        // the widest register tile has no room for a second operand set (it would spill): plain loop, the SIMD
        // partner wave covers the LDS latency
        KStep<RT, NA, EVENK> s0;
        if constexpr (FIRSTSEG) {
            s0.load(xf, xb_, sW, offE, offO, ks);
            s0.template mma<true>(acc_re, acc_im, g, Hf, ks);
            ++ks;
        }
        for (; ks < ks_end; ++ks) {
            s0.load(xf, xb_, sW, offE, offO, ks);
            s0.mma(acc_re, acc_im, g, Hf, ks);
        }
        return;
    }
    KStep<RT, NA, EVENK> s0, s1;
    s0.load(xf, xb_, sW, offE, offO, ks);
    if constexpr (FIRSTSEG) {                            // peeled k-step 0: C = 0
        s1.load(xf, xb_, sW, offE, offO, ks + 1);
        s0.template mma<true>(acc_re, acc_im, g, Hf, ks);
        ++ks;
        if (ks >= ks_end) return;
        s0 = s1;
    }
    for (; ks + 1 < ks_end; ks += 2) {
        s1.load(xf, xb_, sW, offE, offO, ks + 1);
        s0.mma(acc_re, acc_im, g, Hf, ks);
        s0.load(xf, xb_, sW, offE, offO, ks + 2);      // may run one step past the segment: LDS is padded, value unused
        s1.mma(acc_re, acc_im, g, Hf, ks + 1);
    }
    if (ks < ks_end) {
        s0.mma(acc_re, acc_im, g, Hf, ks);
        ++ks;
    }
}

// BWD = false: forward (per-frame partial pooled sums).  BWD = true: the same filterbank recomputation, but the
// epilogue turns the accumulators into dL/dy (pooling + squared-modulus transposes), stores them time-major for the
// tap-gradient GEMM, and accumulates the pooling-width gradient.
template <int RT, int NOFF, bool EVENK, bool BWD>
__global__ __launch_bounds__(kWavesPerWG * 64, kWavesPerWG / 4) void leaf_fused_kernel(const FusedParams p) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int NC = 32 * RT;              // tap columns held by this workgroup: RT Re tiles + RT Im tiles
    const int R = 4 * p.KS;
    float* sW = smem;                        // [R][NC], 16-column halves swapped on odd rows (bank spread)
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    const int li = lane & 15, g = lane >> 4;
    float* xw = smem + (size_t)(R + 4) * NC + (size_t)wave * (p.XS + 16);

    const int tile0 = p.tile_base + blockIdx.y * RT;
    int ks_t[RT];                            // k-steps per tile, non-increasing (columns are sorted by support)
#pragma unroll
    for (int t = 0; t < RT; ++t) ks_t[t] = min(p.KS, __builtin_amdgcn_readfirstlane(p.tile_ks[tile0 + t]));

    // ---- stage this group's taps once per workgroup (only the rows its widest tile needs)
    const int rows_used = 4 * ks_t[0];
    for (int idx = tid; idx < rows_used * NC; idx += kWavesPerWG * 64) {
        const int row = idx / NC, c = idx - row * NC;
        const int tl = c >> 4, j = c & 15;
        const bool is_im = tl >= RT;
        const int src = (is_im ? p.FP : 0) + 16 * (tile0 + (is_im ? tl - RT : tl)) + j;
        sW[row * NC + (c ^ ((row & 1) << 4))] = p.W[(size_t)row * (2 * p.FP) + src];
    }
    __syncthreads();

    // per-lane tap read offsets (floats): row g, 16-col half swap on odd rows
    const int swap = (g & 1) ? 16 : 0;
    const int offE = g * NC + li + swap;     // even local tiles
    const int offO = g * NC + li - swap;     // odd local tiles

    // per-lane base into the pooling table: row = tap column of (tile, li), element = 4g (+ r, + uniform offsets)
    const unsigned goff = (unsigned)((16 * tile0 + li) * p.GJ + 4 * g);

    const int wave_global = blockIdx.x * kWavesPerWG + wave;
    const int wave_stride = gridDim.x * kWavesPerWG;

    // The two waves that share a SIMD (w and w+4) run identical instruction streams; left alone they reach their
    // VALU-only epilogues together and the matrix pipe idles.  Delaying one of them once by about half a unit
    // keeps them out of phase for the rest of the kernel.
    if (wave >= kWavesPerWG / 2 && p.total_tasks > wave_stride)
        for (int i = 0; i < p.desync_sleeps; ++i) __builtin_amdgcn_s_sleep(127);

#if LEAF_TRACE
    int tr_n = 0;
#define LEAF_STAMP()                                                                                     \
    do {                                                                                                 \
        if (blockIdx.x == 0 && blockIdx.y == 0 && lane == 0 && tr_n < 64)                                \
            p.trace[wave * 64 + tr_n] = __builtin_amdgcn_s_memtime();                                    \
        ++tr_n;                                                                                          \
    } while (0)
#else
#define LEAF_STAMP() do { } while (0)
#endif
    bool dma_pending = false;                          // next task's window already streaming into LDS
    float dW[RT];                                      // BWD: running sum of e * dg/ds * grad over this wave's tasks
#pragma unroll
    for (int t = 0; t < RT; ++t) dW[t] = 0.0f;
    for (int task = wave_global; task < p.total_tasks; task += wave_stride) {
        LEAF_STAMP();                                  // task start
        const int b = task / p.nq;
        const int q = p.q_lo + (task - b * p.nq);
        const int n_blk = q * p.hop - p.padL;          // output sample index of the hop-block's first sample
        // ---- stage the waveform window: xw[i] = xz[n_blk - HP + xshift + i]
        if (dma_pending) {
            // the previous task already streamed this window into LDS with direct-to-LDS loads; just wait for them
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            dma_pending = false;
        } else if (!(kAblate & 2)) {
            const float* xb = static_cast<const float*>(p.x) + (size_t)b * p.T;
            const unsigned short* xh = static_cast<const unsigned short*>(p.x) + (size_t)b * p.T;
            const int n0 = n_blk - p.HP + p.xshift;
            for (int i0 = lane; i0 < p.XS; i0 += 4 * 64) {
                float v[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int i = i0 + 64 * j, n = n0 + i;
                    const bool ok = i < p.XS && n >= 0 && n < p.T;
                    if (p.io_bf16)
                        v[j] = ok ? __uint_as_float((unsigned)xh[n] << 16) : 0.0f;
                    else
                        v[j] = ok ? xb[n] : 0.0f;
                }
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    if (i0 + 64 * j < p.XS) xw[i0 + 64 * j] = v[j];
            }
        }
        LEAF_STAMP();                                  // window staged
        // valid output samples of this hop-block (relative index rr): energy outside [0,T) is zero-padded
        const int rr_lo = max(0, -n_blk);
        const int rr_hi = min(p.hop, p.T - n_blk);

        float P[NOFF][RT];                             // forward: per-frame sums; backward: grad of frames q-d
#pragma unroll
        for (int d = 0; d < NOFF; ++d)
#pragma unroll
            for (int t = 0; t < RT; ++t) {
                P[d][t] = 0.0f;
                if constexpr (BWD) {
                    const int m = q - d;
                    if (d < p.noff && m >= 0 && m < p.TP)
                        P[d][t] = p.gcols[((size_t)b * p.TP + m) * p.FP + 16 * (tile0 + t) + li];
                }
            }

        for (int u = 0; u < p.NU; ++u) {
            const int unit_base = 16 * kUB * u;
            if (unit_base >= rr_hi) break;               // nothing of this clip left in the hop-block
            if (unit_base + 16 * kUB <= rr_lo) continue; // unit entirely before the clip starts
            f32x4 acc_re[RT][kUB], acc_im[RT][kUB];       // initialised by k-step 0 (every tile has >= 1 k-step)
            // A operand (signal): lane (row li, k-slot g) of n-block nb reads xw[c0 + 16 nb +- (kk0 + g)]
            const float* xf = xw + p.HP + unit_base + li + g;
            const float* xb_ = xw + p.HP + unit_base + li - g;
            int ks = 0;
            // the wave in its MFMA phase outranks a SIMD partner that is in its epilogue (issue arbitration is by
            // priority, then age): the partner's VALU/VMEM work fills the slots the matrix pipe leaves free.
            LEAF_STAMP();                              // k-loop start
            __builtin_amdgcn_s_setprio(1);
            fused_ksegment<RT, RT, EVENK, true>(acc_re, acc_im, xf, xb_, sW, offE, offO, g, p.Hf, ks, ks_t[RT - 1]);
            if constexpr (RT >= 2)
                fused_ksegment<RT, RT - 1, EVENK>(acc_re, acc_im, xf, xb_, sW, offE, offO, g, p.Hf, ks, ks_t[RT - 2]);
            if constexpr (RT >= 3)
                fused_ksegment<RT, RT - 2, EVENK>(acc_re, acc_im, xf, xb_, sW, offE, offO, g, p.Hf, ks, ks_t[RT - 3]);
            __builtin_amdgcn_s_setprio(0);
            LEAF_STAMP();                              // k-loop end
            if (LEAF_DMA_PREFETCH && u == p.NU - 1 && !p.io_bf16 && !(kAblate & 2)) {
                // This task no longer reads its waveform window: stream the NEXT task's window into the same LDS
                // region with direct-to-LDS loads (no registers), overlapped with this unit's epilogue.  Only for
                // windows that lie entirely inside the clip (edge windows need zero fill -> staged normally).
                const int nt = task + wave_stride;
                if (nt < p.total_tasks) {
                    const int nb_ = nt / p.nq;
                    const int n0n = (p.q_lo + (nt - nb_ * p.nq)) * p.hop - p.padL - p.HP + p.xshift;
                    if (n0n >= 0 && n0n + p.XS <= p.T) {
                        const float* src = static_cast<const float*>(p.x) + (size_t)nb_ * p.T + n0n;
                        for (int i0 = 0; i0 < p.XS; i0 += 64)
                            if (i0 + lane < p.XS)
                                __builtin_amdgcn_global_load_lds(src + i0 + lane, (__attribute__((address_space(3))) void*)(xw + i0), 4, 0, 0);
                        dma_pending = true;
                    }
                }
            }

            // ---- epilogue: |y|^2 times the Gaussian pooling window, accumulated per frame.
            // lane holds, for filter column li of each tile, output samples rr = unit_base + 16 nb + 4g + r, r = 0..3;
            // for frame q-d their pooling taps are j = d*hop + rr .. +3: one 16-byte load from G per (nb, d, tile).
            if (kAblate & 1) {                           // keep the accumulators live, skip the epilogue
#pragma unroll
                for (int t = 0; t < RT; ++t)
#pragma unroll
                    for (int nb = 0; nb < kUB; ++nb) {
                        asm volatile("" ::"v"(acc_re[t][nb]), "v"(acc_im[t][nb]));
                    }
                continue;
            }
            const bool unit_edge = (unit_base < rr_lo) || (unit_base + 16 * kUB > rr_hi);   // clip boundary inside
            if constexpr (BWD) {
                // de[n] = sum_d g[j_d(n)] * grad[q-d]  (transpose of pooling.py:41);  dy = 2 y de  (frontend.py:15-19);
                // d pool_w += e[n] * sum_d (dg/ds)[j_d(n)] * grad[q-d].
                unsigned go = goff;
#pragma unroll
                for (int bi = 0; bi < kUB * RT; ++bi) {
                    const int nb = bi / RT, t = bi % RT;
                    asm volatile("" : "+v"(go), "+v"(dW[t]));
                    f32x4 de = f32x4{0.f, 0.f, 0.f, 0.f}, ds = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int d = 0; d < NOFF; ++d) {
                        const size_t off = (size_t)(16 * t * p.GJ + d * p.hop + unit_base + 16 * nb);
                        const f32x4 gv = *reinterpret_cast<const f32x4u*>((p.G + off) + go);
                        const f32x4 sv = *reinterpret_cast<const f32x4u*>((p.Gs + off) + go);
                        de += gv * P[d][t];
                        ds += sv * P[d][t];
                    }
                    const f32x4 re = acc_re[t][nb], im = acc_im[t][nb];
                    const f32x4 e = re * re + im * im;
                    const int rr0 = unit_base + 16 * nb + 4 * g;
                    float* drow = p.dY + ((size_t)b * p.T + (n_blk + rr0)) * (size_t)(2 * p.FP) + 16 * (tile0 + t) + li;
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const bool in_clip = (rr0 + r >= rr_lo) && (rr0 + r < rr_hi);
                        if (in_clip) {
                            dW[t] = fmaf(e[r], ds[r], dW[t]);
                            drow[(size_t)r * (2 * p.FP)] = 2.0f * re[r] * de[r];
                            drow[(size_t)r * (2 * p.FP) + p.FP] = 2.0f * im[r] * de[r];
                        }
                    }
                }
                (void)unit_edge;
                continue;
            }
            // Software pipeline over the kUB*RT (n-block, tile) batches: the NOFF weight vectors of batch i+1 are in
            // flight while batch i is squared and accumulated.  The table loads do not depend on the MFMA results,
            // so left alone the compiler hoists all of them above the k-loop (180 registers -> spills); an opaque
            // asm re-defining the lane offset (and touching the running sums) pins each batch in program order.
            f32x4 gwb[2][NOFF];
            auto load_batch = [&](f32x4 (&dst)[NOFF], int nb, int t, unsigned go) {
#pragma unroll
                for (int d = 0; d < NOFF; ++d)     // uniform (SGPR) base + one per-lane 32-bit offset
                    dst[d] = *reinterpret_cast<const f32x4u*>(
                        (p.G + (size_t)(16 * t * p.GJ + d * p.hop + unit_base + 16 * nb)) + go);
            };
            unsigned go = goff;
            asm volatile("" : "+v"(go));
            load_batch(gwb[0], 0, 0, go);
#pragma unroll
            for (int bi = 0; bi < kUB * RT; ++bi) {
                const int nb = bi / RT, t = bi % RT;
                if (bi + 1 < kUB * RT) load_batch(gwb[(bi + 1) & 1], (bi + 1) / RT, (bi + 1) % RT, go);
                f32x4 e = acc_re[t][nb] * acc_re[t][nb] + acc_im[t][nb] * acc_im[t][nb];
                if (unit_edge) {                         // energy outside [0,T) is zero-padded (pooling.py:37)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int rr = unit_base + 16 * nb + 4 * g + r;
                        e[r] = ((rr >= rr_lo) && (rr < rr_hi)) ? e[r] : 0.0f;
                    }
                }
#pragma unroll
                for (int d = 0; d < NOFF; ++d)
#pragma unroll
                    for (int r = 0; r < 4; ++r) P[d][t] = fmaf(e[r], gwb[bi & 1][d][r], P[d][t]);
                asm volatile("" : "+v"(go), "+v"(P[0][t]));
            }
            LEAF_STAMP();                              // epilogue end
        }
        if constexpr (BWD) continue;
        // ---- reduce the 4 k-slot groups (same filter column, different samples) and store partials
#pragma unroll
        for (int d = 0; d < NOFF; ++d) {
            const int m = q - d;
#pragma unroll
            for (int t = 0; t < RT; ++t) {
                float v = P[d][t];
                v += __shfl_xor(v, 16);
                v += __shfl_xor(v, 32);
                if (!(kAblate & 4) && g == 0 && d < p.noff && m >= 0 && m < p.TP)
                    p.part[(((size_t)b * p.TP + m) * p.noff + d) * p.FP + 16 * (tile0 + t) + li] = v;
            }
        }
    }
    if constexpr (BWD) {
#pragma unroll
        for (int t = 0; t < RT; ++t) {
            float v = dW[t];
            v += __shfl_xor(v, 16);
            v += __shfl_xor(v, 32);
            if (g == 0) p.dwpart[(size_t)wave_global * p.FP + 16 * (tile0 + t) + li] = v;
        }
    }
}

// Sum the partials of every frame, add bias, floor (frontend.py:84), then the EMA recurrence and PCEN
// (postprocessing.py:13-28, 62-69).  One workgroup per clip, 64-frame chunks:
//   phase 1  all threads: pooled[f][m] -> LDS (partial reads coalesced across filters)
//   phase 2  one wave per filter, lanes = frames: the first-order recurrence M_m = w p_m + (1-w) M_{m-1} is an
//            affine map composition, scanned across the wavefront with 6 shuffle steps and a carried state;
//            PCEN is applied pointwise and rows are written with 256-byte coalesced stores.
// mode bit0: PCEN, bit1: log1p (extension)
constexpr int kFinThreads = 1024;
constexpr int kFinPer = 4;        // pooled values a thread gathers per pass (independent loads in flight)
__global__ __launch_bounds__(kFinThreads) void finalize_kernel(
    const float* __restrict__ part, int F, int FP, int TP, int noff, int q_lo, int q_hi, const int* __restrict__ col_of,
    const float* __restrict__ bias, const float* __restrict__ alpha, const float* __restrict__ delta,
    const float* __restrict__ root, const float* __restrict__ ema_w, float floor_, int mode, void* __restrict__ out_,
    float* __restrict__ raw_out /* optional [B][F][TP]: bias + pooled sum before the floor (saved for backward) */) {
    extern __shared__ __attribute__((aligned(16))) float fsm[];
    float* out = static_cast<float*>(out_);
    unsigned short* outh = static_cast<unsigned short*>(out_);
    float* sv = fsm;                 // [F][65] pooled values of the current 64-frame chunk
    float* scarry = sv + F * 65;     // [F] EMA state carried across chunks
    float* s_dr = scarry + F;        // [F] delta^(1/r)
    int* s_col = reinterpret_cast<int*>(s_dr + F);   // [F] tap column of each filter
    const int b = blockIdx.x, tid = threadIdx.x;
    const int wave = tid >> 6, lane = tid & 63;
    for (int f = tid; f < F; f += kFinThreads) {
        s_col[f] = col_of[f];
        s_dr[f] = (mode & 1) ? powf(delta[f], 1.0f / fmaxf(root[f], 1.0f)) : 0.0f;
    }
    __syncthreads();
    for (int m0 = 0; m0 < TP; m0 += 64) {
        const int nm = min(64, TP - m0);
        for (int base = 0; base < nm * F; base += kFinThreads * kFinPer) {
            float acc[kFinPer];
            int slot[kFinPer];
#pragma unroll
            for (int i = 0; i < kFinPer; ++i) {
                const int idx = base + i * kFinThreads + tid;
                acc[i] = 0.0f;
                slot[i] = -1;
                if (idx < nm * F) {
                    const int mm = idx / F, f = idx - mm * F;
                    const int m = m0 + mm;
                    const float* pp = part + (((size_t)b * TP + m) * noff) * FP + s_col[f];
                    for (int dd = 0; dd < noff; ++dd) {
                        const int q = m + dd;
                        if (q >= q_lo && q <= q_hi) acc[i] += pp[(size_t)dd * FP];
                    }
                    acc[i] += bias ? bias[f] : 0.0f;
                    slot[i] = f * 65 + mm;
                }
            }
#pragma unroll
            for (int i = 0; i < kFinPer; ++i)
                if (slot[i] >= 0) {
                    sv[slot[i]] = (mode & 8) ? acc[i] : fmaxf(acc[i], kPooledFloor);
                    if (raw_out) {
                        const int f = slot[i] / 65, mm = slot[i] - f * 65;
                        raw_out[((size_t)b * F + f) * TP + m0 + mm] = acc[i];
                    }
                }
        }
        __syncthreads();
        for (int f = wave; f < F; f += kFinThreads / 64) {
            const float v = lane < nm ? sv[f * 65 + lane] : 0.0f;
            float r = v;
            if (mode & 8) {                              // backward: pre-floor pooled value
            } else if (mode & 1) {
                const float w = fminf(fmaxf(ema_w[f], 0.0f), 1.0f);
                float A = lane < nm ? 1.0f - w : 1.0f;       // M_m = A_m * M_{m-1} + Bv_m
                float Bv = lane < nm ? w * v : 0.0f;
#pragma unroll
                for (int off = 1; off < 64; off <<= 1) {
                    const float Ap = __shfl_up(A, off), Bp = __shfl_up(Bv, off);
                    if (lane >= off) {
                        Bv = fmaf(A, Bp, Bv);
                        A *= Ap;
                    }
                }
                const float carry = (m0 == 0) ? sv[f * 65] : scarry[f];   // state starts at p_0 (postprocessing.py:15)
                const float M = fmaf(A, carry, Bv);
                const float last = __shfl(M, nm - 1);
                if (lane == 0) scarry[f] = last;
                const float a = fminf(alpha[f], 1.0f);
                const float inv_r = 1.0f / fmaxf(root[f], 1.0f);
                // (floor+M)^a through accurate log2f/exp2f (its error is damped by the outer root); the outer
                // power feeds a cancelling subtraction and keeps the full-accuracy powf.
                const float den = exp2f(a * log2f(floor_ + M));
                r = powf(v / den + delta[f], inv_r) - s_dr[f];
            } else if (mode & 2) {
                r = log1pf(v);
            }
            if (lane < nm) {
                const size_t o = ((size_t)b * F + f) * TP + m0 + lane;
                if (mode & 4) {                              // bf16 output, round to nearest even
                    const unsigned u = __float_as_uint(r);
                    outh[o] = (unsigned short)((u + 0x7fffu + ((u >> 16) & 1u)) >> 16);
                } else {
                    out[o] = r;
                }
            }
        }
        __syncthreads();
    }
}

// floor + optional log1p on an already pooled (B,F,T') tensor (staged path without PCEN)
__global__ void floor_kernel(const float* __restrict__ p, size_t n, int mode, float* __restrict__ out) {
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= n) return;
    const float v = fmaxf(p[idx], kPooledFloor);
    out[idx] = (mode & 2) ? log1pf(v) : v;
}

// ---------------------------------------------------------------------------------------------
// backward (staged, correctness-first): gradients of a scalar loss w.r.t. the seven parameters (and
// optionally x) given dL/d out.  Every forward intermediate is recomputed on the device with the staged
// kernels above; nothing is kept from the forward call.  Mirrors what autograd derives for the reference
// graph (frontend.py:78-89), including its clamp sub-gradients:
//   torch.clamp  -> gradient passes where lo <= x <= hi          (convolution.py:19-20, impulse_responses.py:75,
//                                                                  postprocessing.py:14)
//   torch.min/max against a scalar tensor -> the selected side; an exact tie splits 1/2 (postprocessing.py:63-64)
//   torch.maximum(p, 1e-5)               -> passes where p > 1e-5 (frontend.py:84)
// ---------------------------------------------------------------------------------------------

// One lane per (b,f) row.  raw = pooled before the floor.  Forward EMA is recomputed into `ema`, then the
// reverse-time sweep produces g_pre (grad w.r.t. raw) and the row's contributions to d alpha, d delta, d root,
// d ema_w in rowsum[row][4].  mode bit0: PCEN on.
__global__ void pcen_bwd_rows_kernel(const float* __restrict__ raw, const float* __restrict__ gout, int BF, int F, int TP,
                                     const float* __restrict__ alpha, const float* __restrict__ delta,
                                     const float* __restrict__ root, const float* __restrict__ ema_w, float floor_,
                                     int mode, float* __restrict__ ema, float* __restrict__ gpre,
                                     float* __restrict__ rowsum, const int* __restrict__ col_of, int FP,
                                     float* __restrict__ gcols) {
    const int row = blockIdx.x * blockDim.x + threadIdx.x;
    if (row >= BF) return;
    const float* r = raw + (size_t)row * TP;
    const float* go = gout + (size_t)row * TP;
    float* gp = gpre + (size_t)row * TP;
    const int f = row % F;
    // fused backward: also a [B][TP][FP] copy with filters in tap-column order
    float* gc = gcols ? gcols + (size_t)(row / F) * TP * FP + col_of[f] : nullptr;
    if (!(mode & 1)) {
        for (int m = 0; m < TP; ++m) {
            const float v = r[m] > kPooledFloor ? go[m] : 0.0f;
            gp[m] = v;
            if (gc) gc[(size_t)m * FP] = v;
        }
        return;
    }
    float* M = ema + (size_t)row * TP;
    const float w = fminf(fmaxf(ema_w[f], 0.0f), 1.0f), omw = 1.0f - w;
    const float a = fminf(alpha[f], 1.0f);
    const float reff = fmaxf(root[f], 1.0f), rho = 1.0f / reff;
    const float d = delta[f];
    const float d_rho = powf(d, rho), ln_d = logf(d);
    float state = fmaxf(r[0], kPooledFloor);
    for (int m = 0; m < TP; ++m) {
        const float p = fmaxf(r[m], kPooledFloor);
        state = w * p + omw * state;
        M[m] = state;
    }
    float s_a = 0.f, s_d = 0.f, s_rho = 0.f, s_w = 0.f, gM_next = 0.f;
    const float p0 = fmaxf(r[0], kPooledFloor);
    for (int m = TP - 1; m >= 0; --m) {
        const float p = fmaxf(r[m], kPooledFloor);
        const float Mf = floor_ + M[m];
        const float u = powf(Mf, a);
        const float v = p / u + d;
        const float vr = powf(v, rho);
        const float g = go[m];
        const float dv = rho * vr / v * g;
        s_d += dv - rho * d_rho / d * g;
        s_rho += (vr * logf(v) - d_rho * ln_d) * g;
        float dp = dv / u;
        const float du = -dv * p / (u * u);
        s_a += du * u * logf(Mf);
        const float gM = du * a * u / Mf + omw * gM_next;
        dp += w * gM;
        const float Mprev = m > 0 ? M[m - 1] : p0;
        s_w += gM * (p - Mprev);
        if (m == 0) dp += omw * gM;                 // the recurrence starts from p_0 (postprocessing.py:15)
        gM_next = gM;
        const float gv = r[m] > kPooledFloor ? dp : 0.0f;
        gp[m] = gv;
        if (gc) gc[(size_t)m * FP] = gv;
    }
    const float al = alpha[f], ro = root[f], ew = ema_w[f];
    float* rs = rowsum + (size_t)row * 4;
    rs[0] = al < 1.0f ? s_a : (al == 1.0f ? 0.5f * s_a : 0.0f);
    rs[1] = s_d;
    const float g_reff = -s_rho * rho * rho;
    rs[2] = ro > 1.0f ? g_reff : (ro == 1.0f ? 0.5f * g_reff : 0.0f);
    rs[3] = (ew >= 0.0f && ew <= 1.0f) ? s_w : 0.0f;
}

// d e[b,f,n] = sum_m g[f][n + padL - m hop] * gpre[b,f,m]  (transpose of pooling.py:41), then
// dy[b,2f,n] = 2 y_re de, dy[b,2f+1,n] = 2 y_im de written over y  (frontend.py:15-19).
__global__ void pool_bwd_dy_kernel(float* __restrict__ y, const float* __restrict__ g, const float* __restrict__ gpre,
                                   int F, int T, int TP, int K, int hop, int padL) {
    const int n = blockIdx.x * blockDim.x + threadIdx.x;
    const int f = blockIdx.y, b = blockIdx.z;
    if (n >= T) return;
    const float* w = g + (size_t)f * K;
    const float* gp = gpre + ((size_t)b * F + f) * TP;
    const int np = n + padL;
    const int m_hi = min(TP - 1, np / hop);
    const int m_lo = max(0, (np - K + hop) / hop);          // smallest m with np - m*hop <= K-1
    float de = 0.0f;
    for (int m = m_lo; m <= m_hi; ++m) {
        const int j = np - m * hop;
        if (j >= 0 && j < K) de = fmaf(w[j], gp[m], de);
    }
    const size_t ire = ((size_t)b * 2 * F + 2 * f) * T + n;
    y[ire] *= 2.0f * de;
    y[ire + T] *= 2.0f * de;
}

// dg[f][j] = sum_{b,m} gpre[b,f,m] * ez[b,f,m hop + j - padL]
__global__ void pool_bwd_dg_kernel(const float* __restrict__ e, const float* __restrict__ gpre, int B, int F, int T, int TP,
                                   int K, int hop, int padL, float* __restrict__ dg) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    const int f = blockIdx.y;
    if (j >= K) return;
    float acc = 0.0f;
    for (int b = 0; b < B; ++b) {
        const float* eb = e + ((size_t)b * F + f) * T;
        const float* gp = gpre + ((size_t)b * F + f) * TP;
        for (int m = 0; m < TP; ++m) {
            const int n = m * hop + j - padL;
            if (n >= 0 && n < T) acc = fmaf(gp[m], eb[n], acc);
        }
    }
    dg[(size_t)f * K + j] = acc;
}

// One block per filter: d pool_b, d pool_w and the PCEN parameter sums over the batch.
__global__ void param_reduce_kernel(const float* __restrict__ gpre, const float* __restrict__ dg,
                                    const float* __restrict__ g, const float* __restrict__ rowsum,
                                    const float* __restrict__ pool_w, int B, int F, int TP, int K, int mode,
                                    const float* __restrict__ dwpart, int dw_rows, int FP,
                                    const int* __restrict__ col_of, float* __restrict__ g_pool_w, float* __restrict__ g_pool_b, float* __restrict__ g_alpha,
                                    float* __restrict__ g_delta, float* __restrict__ g_root, float* __restrict__ g_ema) {
    __shared__ float red[256];
    const int f = blockIdx.x, tid = threadIdx.x;
    auto block_sum = [&](float v) {
        red[tid] = v;
        __syncthreads();
        for (int s = 128; s > 0; s >>= 1) {
            if (tid < s) red[tid] += red[tid + s];
            __syncthreads();
        }
        const float r = red[0];
        __syncthreads();
        return r;
    };
    float acc = 0.0f;
    for (int i = tid; i < B * TP; i += 256) {
        const int b = i / TP, m = i - b * TP;
        acc += gpre[((size_t)b * F + f) * TP + m];
    }
    const float sb = block_sum(acc);
    // d g/d s = g * (j - c)^2 / (c^2 s^3), c = (K-1)/2   (impulse_responses.py:75-80)
    const float wr = pool_w[f];
    const float sig = pool_sigma(wr, K);
    const float c = 0.5f * (float)(K - 1);
    acc = 0.0f;
    if (dwpart) {                                     // fused backward: per-wave partial sums, tap-column order
        const int col = col_of[f];
        for (int i = tid; i < dw_rows; i += 256) acc += dwpart[(size_t)i * FP + col];
    } else {
        for (int j = tid; j < K; j += 256) {
            const float t = (float)j - c;
            acc += dg[(size_t)f * K + j] * g[(size_t)f * K + j] * (t * t) / (c * c * sig * sig * sig);
        }
    }
    const float sw = block_sum(acc);
    float sums[4] = {0.f, 0.f, 0.f, 0.f};
    if (mode & 1) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            acc = 0.0f;
            for (int b = tid; b < B; b += 256) acc += rowsum[((size_t)b * F + f) * 4 + q];
            sums[q] = block_sum(acc);
        }
    }
    if (tid == 0) {
        if (g_pool_b) g_pool_b[f] = sb;
        g_pool_w[f] = (wr >= 2.0f / (float)K && wr <= 0.5f) ? sw : 0.0f;
        if (mode & 1) {
            g_alpha[f] = sums[0];
            g_delta[f] = sums[1];
            g_root[f] = sums[2];
            g_ema[f] = sums[3];
        }
    }
}

// dtaps partial per clip: part[b][c][j] = sum_n dy[b,c,n] * xz[b, n + j - padL]   (transpose of convolution.py:97 w.r.t. weights)
__global__ void dtaps_partial_kernel(const float* __restrict__ dy, const float* __restrict__ x, int T, int C, int K, int padL,
                                     float* __restrict__ part) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    const int c = blockIdx.y, b = blockIdx.z;
    if (j >= K) return;
    const float* d = dy + ((size_t)b * C + c) * T;
    const float* xb = x + (size_t)b * T;
    const int off = j - padL;
    const int n0 = max(0, -off), n1 = min(T, T - off);
    float acc = 0.0f;
    for (int n = n0; n < n1; ++n) acc = fmaf(d[n], xb[n + off], acc);
    part[((size_t)b * C + c) * K + j] = acc;
}

// One block per filter: sum the per-clip tap gradients over the batch and chain them through the Gabor formula
// (impulse_responses.py:5-16) to (mu, sigma):  d hr/d mu = -t hi, d hi/d mu = t hr, d h/d sigma = h (t^2/s^3 - 1/s).
__global__ void dkernel_kernel(const float* __restrict__ part, const float* __restrict__ taps,
                               const float* __restrict__ kernel, int B, int F, int K, GaborBounds bd,
                               float* __restrict__ g_kernel) {
    __shared__ float red[256];
    const int f = blockIdx.x, tid = threadIdx.x;
    const float mu_raw = kernel[2 * f], sg_raw = kernel[2 * f + 1];
    const float sg = fminf(fmaxf(sg_raw, bd.sigma_lo), bd.sigma_hi);
    float a_mu = 0.0f, a_sg = 0.0f;
    for (int j = tid; j < K; j += 256) {
        float dre = 0.0f, dim = 0.0f;
        for (int b = 0; b < B; ++b) {
            dre += part[((size_t)b * 2 * F + 2 * f) * K + j];
            dim += part[((size_t)b * 2 * F + 2 * f + 1) * K + j];
        }
        const float t = (float)(j - K / 2);
        const float hr = taps[(size_t)(2 * f) * K + j], hi = taps[(size_t)(2 * f + 1) * K + j];
        a_mu += t * (dim * hr - dre * hi);
        a_sg += (dre * hr + dim * hi) * (t * t / (sg * sg * sg) - 1.0f / sg);
    }
    float out2[2];
    float vals[2] = {a_mu, a_sg};
    for (int q = 0; q < 2; ++q) {
        red[tid] = vals[q];
        __syncthreads();
        for (int s = 128; s > 0; s >>= 1) {
            if (tid < s) red[tid] += red[tid + s];
            __syncthreads();
        }
        out2[q] = red[0];
        __syncthreads();
    }
    if (tid == 0) {
        g_kernel[2 * f] = (mu_raw >= 0.0f && mu_raw <= 3.14159274101257324f) ? out2[0] : 0.0f;
        g_kernel[2 * f + 1] = (sg_raw >= bd.sigma_lo && sg_raw <= bd.sigma_hi) ? out2[1] : 0.0f;
    }
}

// dx[b,i] = sum_c sum_j taps[c][j] * dy[b,c,i - j + padL]
__global__ void dx_kernel(const float* __restrict__ dy, const float* __restrict__ taps, int T, int C, int K, int padL,
                          float* __restrict__ dx) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const int b = blockIdx.y;
    if (i >= T) return;
    float acc = 0.0f;
    for (int c = 0; c < C; ++c) {
        const float* d = dy + ((size_t)b * C + c) * T;
        const float* w = taps + (size_t)c * K;
        const int j0 = max(0, i + padL - (T - 1)), j1 = min(K, i + padL + 1);
        for (int j = j0; j < j1; ++j) acc = fmaf(w[j], d[i + padL - j], acc);
    }
    dx[(size_t)b * T + i] = acc;
}

// ---------------------------------------------------------------------------------------------
// fused backward, phase C: tap gradients as an fp32-MFMA GEMM.
//   dH[kk][c] = sum_{b,n} S_kk[b,n] * dY[b,n][c]   (c < FP, Re columns)      S_kk[n] = x[n+kk] + x[n-kk]
//   dH[kk][c] = sum_{b,n} D_kk[b,n] * dY[b,n][c]   (c >= FP, Im columns)     D_kk[n] = x[n+kk] - x[n-kk]
// i.e. the transpose of the forward GEMMs w.r.t. the tap table W, with the same Hermitian operands built from an
// LDS waveform window.  Rows = 16 tap rows per wave (one k-tile each), columns = the group's 16-filter tiles,
// reduction = time.  A workgroup walks 64-sample chunks (waveform window + dY tile double-buffered in LDS, next
// chunk prefetched into registers under the MFMAs) and finally writes its partial dH; a small kernel sums the
// partials and chains them to (mu, sigma).
// ---------------------------------------------------------------------------------------------
struct DtapsParams {
    const float* x;        // [B][T]
    const float* dY;       // [B*T][2*FP]
    const int* tile_ks;    // k-steps per column tile (support-sorted)
    float* dHpart;         // [gridDim.x][16*NKT][2*FP]
    int B, T, FP, K, Hf, xshift;
    int NKT;               // 16-row k-tiles
    int NW;                // waves per workgroup
    int NS;                // samples per chunk (multiple of 16)
    int HPc;               // window halo = 16*NKT
    int XSC;               // window floats = NS + 2*HPc
    int LD;                // LDS row stride of the dY tile = 2*FP + 16
    int nch;               // chunks per clip
    int total_chunks;      // B * nch
    int tile_base;         // first column tile of this launch's group 0
};

template <int RT, int NA, bool EVENK>
__device__ __forceinline__ void dtaps_ktile(f32x4 (&acc)[2 * RT], const float* xc, const float* sdy, int LD, int colre,
                                            int colim, int krow, int g, int Hf, int NS) {
    for (int nb = 0; nb < NS / 16; ++nb) {
#pragma unroll
        for (int s4 = 0; s4 < 4; ++s4) {
            const int rr = 16 * nb + 4 * s4 + g;
            float fw = xc[rr + krow];
            const float bw = xc[rr - krow];
            if (EVENK) fw = krow <= Hf ? fw : 0.0f;
            const float sv = fw + bw, dv = fw - bw;
            const float* row = sdy + rr * LD;
#pragma unroll
            for (int t = 0; t < NA; ++t) {
                acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(sv, row[colre + 16 * t], acc[t], 0, 0, 0);
                acc[RT + t] = __builtin_amdgcn_mfma_f32_16x16x4f32(dv, row[colim + 16 * t], acc[RT + t], 0, 0, 0);
            }
        }
    }
}

constexpr int kDtPF = 4;      // float4 registers per thread for the dY tile prefetch
template <int RT, int TPW, bool EVENK>
__global__ __launch_bounds__(1024) void dtaps_mfma_kernel(const DtapsParams p) {
    extern __shared__ __attribute__((aligned(16))) float dsm[];
    const int tid = threadIdx.x, nthreads = blockDim.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    const int li = lane & 15, g = lane >> 4;
    const int tile_floats = p.NS * p.LD;
    const int buf_floats = (p.XSC + 3) / 4 * 4 + tile_floats;
    const int tile0 = p.tile_base + blockIdx.y * RT;
    const int colre = 16 * tile0 + li, colim = p.FP + 16 * tile0 + li;
    const int row4 = 2 * p.FP / 4;                       // float4 per dY row
    const int n4 = p.NS * row4;                          // float4 per dY tile

    // k-tile -> (wave, slot) assignment.  Column tiles are support-sorted, so low k-tiles carry more MFMAs (all column
    // tiles reach them) than high ones: a round-robin split leaves one SIMD with ~30 % more work.  Thread 0 does a
    // longest-processing-time greedy that balances the four SIMDs (waves w, w+4, .. share SIMD w & 3).
    __shared__ int s_kt[16 * 3];
    if (tid == 0) {
        int load[16], cnt[16], simd_load[4] = {0, 0, 0, 0};
        for (int w = 0; w < 16; ++w) load[w] = cnt[w] = 0;
        for (int i = 0; i < 16 * 3; ++i) s_kt[i] = -1;
        for (int kt = 0; kt < p.NKT; ++kt) {
            int work = 0;
            for (int t = 0; t < RT; ++t) work += (4 * p.tile_ks[tile0 + t] > 16 * kt) ? 1 : 0;
            if (work == 0) continue;
            int best = -1, best_key = 1 << 30;
            for (int w = 0; w < p.NW; ++w) {
                if (cnt[w] >= TPW) continue;
                const int key = simd_load[w & 3] * 64 + load[w];
                if (key < best_key) { best_key = key; best = w; }
            }
            s_kt[best * TPW + cnt[best]] = kt;
            cnt[best]++; load[best] += work; simd_load[best & 3] += work;
        }
    }
    __syncthreads();
    int na[TPW], ktile[TPW];                              // owned k-tiles and their active column tiles
#pragma unroll
    for (int tp = 0; tp < TPW; ++tp) {
        const int kt = __builtin_amdgcn_readfirstlane(s_kt[wave * TPW + tp]);
        ktile[tp] = kt;
        int n = 0;
        for (int t = 0; t < RT; ++t) n += (kt >= 0 && 4 * p.tile_ks[tile0 + t] > 16 * kt) ? 1 : 0;
        na[tp] = __builtin_amdgcn_readfirstlane(n);
    }
    f32x4 acc[TPW][2 * RT];
#pragma unroll
    for (int tp = 0; tp < TPW; ++tp)
#pragma unroll
        for (int c = 0; c < 2 * RT; ++c) acc[tp][c] = f32x4{0.f, 0.f, 0.f, 0.f};

    f32x4 pre[kDtPF];
    float prex[2];
    auto load_chunk = [&](int chunk) {
        const int b = chunk / p.nch, n0 = (chunk - b * p.nch) * p.NS;
        const float* src = p.dY + ((size_t)b * p.T + n0) * (size_t)(2 * p.FP);
#pragma unroll
        for (int i = 0; i < kDtPF; ++i) {
            const int idx = tid + i * nthreads;
            pre[i] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (idx < n4 && n0 + idx / row4 < p.T) pre[i] = *reinterpret_cast<const f32x4*>(src + (size_t)idx * 4);
        }
        const float* xb = p.x + (size_t)b * p.T;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int idx = tid + i * nthreads;
            const int n = n0 - p.HPc + p.xshift + idx;
            prex[i] = (idx < p.XSC && n >= 0 && n < p.T) ? xb[n] : 0.0f;
        }
    };
    auto store_chunk = [&](float* buf) {
        float* xw = buf;
        float* sdy = buf + (p.XSC + 3) / 4 * 4;
#pragma unroll
        for (int i = 0; i < kDtPF; ++i) {
            const int idx = tid + i * nthreads;
            if (idx < n4) {
                const int row = idx / row4, c4 = idx - row * row4;
                *reinterpret_cast<f32x4*>(sdy + row * p.LD + 4 * c4) = pre[i];
            }
        }
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int idx = tid + i * nthreads;
            if (idx < p.XSC) xw[idx] = prex[i];
        }
    };

    int chunk = blockIdx.x;
    int cur = 0;
    if (chunk < p.total_chunks) {
        load_chunk(chunk);
        store_chunk(dsm);
    }
    __syncthreads();
    for (; chunk < p.total_chunks; chunk += gridDim.x) {
        const int next = chunk + gridDim.x;
        if (next < p.total_chunks) load_chunk(next);
        const float* buf = dsm + (size_t)cur * buf_floats;
        const float* xc = buf + p.HPc;
        const float* sdy = buf + (p.XSC + 3) / 4 * 4;
#pragma unroll
        for (int tp = 0; tp < TPW; ++tp) {
            const int krow = 16 * ktile[tp] + li;
            if (na[tp] == RT) dtaps_ktile<RT, RT, EVENK>(acc[tp], xc, sdy, p.LD, colre, colim, krow, g, p.Hf, p.NS);
            if constexpr (RT >= 2)
                if (na[tp] == RT - 1)
                    dtaps_ktile<RT, RT - 1, EVENK>(acc[tp], xc, sdy, p.LD, colre, colim, krow, g, p.Hf, p.NS);
            if constexpr (RT >= 3)
                if (na[tp] == RT - 2)
                    dtaps_ktile<RT, RT - 2, EVENK>(acc[tp], xc, sdy, p.LD, colre, colim, krow, g, p.Hf, p.NS);
        }
        if (next < p.total_chunks) store_chunk(dsm + (size_t)(cur ^ 1) * buf_floats);
        __syncthreads();
        cur ^= 1;
    }
    // partial dH of this workgroup: D layout row = 4g + r (tap row within the k-tile), col = li
    float* outp = p.dHpart + (size_t)blockIdx.x * (16 * p.NKT) * (2 * p.FP);
#pragma unroll
    for (int tp = 0; tp < TPW; ++tp) {
        const int kt = ktile[tp];
        if (kt < 0) continue;
#pragma unroll
        for (int t = 0; t < RT; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const size_t rowoff = (size_t)(16 * kt + 4 * g + r) * (2 * p.FP);
                outp[rowoff + colre + 16 * t] = acc[tp][t][r];
                outp[rowoff + colim + 16 * t] = acc[tp][RT + t][r];
            }
    }
}

// sum the per-workgroup partial dH slabs: out[i] = sum_w part[w][i]
__global__ void dh_reduce_kernel(const float* __restrict__ part, int nparts, size_t n, float* __restrict__ out) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float acc = 0.0f;
    for (int w = 0; w < nparts; ++w) acc += part[(size_t)w * n + i];
    out[i] = acc;
}

// One block per filter: sum the workgroup partials of dH and chain through the Gabor formula using the tap table
// itself (W = h * scale, and the scale cancels): d mu = sum_kk kk (dH_im W_re - dH_re W_im),
// d sigma = sum_kk (dH_re W_re + dH_im W_im) (kk^2/s^3 - 1/s); clamp sub-gradients as torch.clamp.
__global__ void dkernel_fused_kernel(const float* __restrict__ dHpart, int nparts, int Rp, const float* __restrict__ W,
                                     int R, int FP, const int* __restrict__ col_of, const float* __restrict__ kernel,
                                     int F, GaborBounds bd, float* __restrict__ g_kernel) {
    __shared__ float red[256];
    const int f = blockIdx.x, tid = threadIdx.x;
    const int c = col_of[f];
    const float mu_raw = kernel[2 * f], sg_raw = kernel[2 * f + 1];
    const float sg = fminf(fmaxf(sg_raw, bd.sigma_lo), bd.sigma_hi);
    float a_mu = 0.0f, a_sg = 0.0f;
    for (int kk = tid; kk < R; kk += 256) {
        float dre = 0.0f, dim = 0.0f;
        for (int w = 0; w < nparts; ++w) {
            const float* row = dHpart + ((size_t)w * Rp + kk) * (2 * FP);
            dre += row[c];
            dim += row[FP + c];
        }
        const float wre = W[(size_t)kk * (2 * FP) + c], wim = W[(size_t)kk * (2 * FP) + FP + c];
        const float t = (float)kk;
        a_mu += t * (dim * wre - dre * wim);
        a_sg += (dre * wre + dim * wim) * (t * t / (sg * sg * sg) - 1.0f / sg);
    }
    float res[2];
    const float vals[2] = {a_mu, a_sg};
    for (int q = 0; q < 2; ++q) {
        red[tid] = vals[q];
        __syncthreads();
        for (int s2 = 128; s2 > 0; s2 >>= 1) {
            if (tid < s2) red[tid] += red[tid + s2];
            __syncthreads();
        }
        res[q] = red[0];
        __syncthreads();
    }
    if (tid == 0) {
        g_kernel[2 * f] = (mu_raw >= 0.0f && mu_raw <= 3.14159274101257324f) ? res[0] : 0.0f;
        g_kernel[2 * f + 1] = (sg_raw >= bd.sigma_lo && sg_raw <= bd.sigma_hi) ? res[1] : 0.0f;
    }
}

// ---------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------

inline int ceil_div(int a, int b) { return (a + b - 1) / b; }
inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

int num_cus() {
    static int cached = 0;
    if (cached > 0) return cached;
    int dev = 0, n = 0;
    if (hipGetDevice(&dev) != hipSuccess) return 256;
    if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
    cached = n;
    return n;
}

struct FusedPlan {
    bool ok;
    int FP, ntiles, KS, R, Hf, xshift, NBH, NU, HP, XS, q_lo, q_hi, nq, noff, noff_t, padL, TP;
    int rt_main, groups_main, rt_rem;
    int GJ;
    size_t w_floats, g_floats, part_floats, meta_ints;
};

// +4 tap rows and +16 window floats per wave: the k-loop prefetches one step past its last k-step
inline size_t fused_lds_bytes(int R, int rt, int XS) {
    return ((size_t)(R + 4) * 32 * rt + (size_t)kWavesPerWG * (XS + 16)) * 4;
}

FusedPlan make_plan(int B, int T, int F, int K, int hop) {
    FusedPlan pl{};
    pl.padL = K / 2 + K % 2 - 1;
    pl.TP = (T + (K - 1) - K) / hop + 1;
    pl.FP = 16 * ceil_div(F, 16);
    pl.ntiles = pl.FP / 16;
    pl.KS = ceil_div(K / 2 + 1, 4);
    pl.R = 4 * pl.KS;
    pl.Hf = (K - 1) / 2;
    pl.xshift = K / 2 - pl.padL;
    pl.NBH = ceil_div(hop, 16);
    pl.NU = ceil_div(pl.NBH, kUB);
    pl.HP = 4 * pl.KS;
    pl.XS = 16 * kUB * pl.NU + 2 * pl.HP;
    pl.q_lo = pl.padL / hop;
    pl.q_hi = (T - 1 + pl.padL) / hop;
    pl.nq = pl.q_hi - pl.q_lo + 1;
    pl.noff = (K - 1) / hop + 1;
    pl.noff_t = pl.noff <= 1 ? 1 : (pl.noff <= 3 ? 3 : (pl.noff <= 6 ? 6 : 0));
    pl.w_floats = (size_t)pl.R * 2 * pl.FP;
    pl.GJ = ((std::max(pl.noff_t, 1) - 1) * hop + 16 * kUB * pl.NU + 3) / 4 * 4;
    pl.g_floats = (size_t)pl.FP * pl.GJ;
    pl.part_floats = (size_t)B * pl.TP * pl.noff * pl.FP;
    pl.meta_ints = (size_t)2 * pl.FP + pl.ntiles;          // perm[FP], col_of[FP], tile_ks[ntiles]
    pl.rt_main = 0;
    // Widest register tile whose taps fit LDS -- unless the batch is so small that (tasks x filter groups) would not
    // even give every wave slot of the chip one task: then narrower tiles (more filter groups, shorter per-wave
    // dependency chains) cut the latency of a small forward.
    const long long wave_slots = (long long)num_cus() * kWavesPerWG;
    for (int rt = 3; rt >= 1; --rt) {
        if (rt > pl.ntiles || fused_lds_bytes(pl.R, rt, pl.XS) > (size_t)kMaxLds) continue;
        if (pl.rt_main == 0) pl.rt_main = rt;
        if ((long long)B * pl.nq * ceil_div(pl.ntiles, rt) >= wave_slots) break;
        pl.rt_main = rt;
    }
    pl.ok = pl.rt_main > 0 && pl.noff_t > 0 && pl.FP <= kMaxFP && (long long)B * pl.nq < (1ll << 30) &&
            (double)pl.part_floats < 2.0e9;
    if (pl.ok) {
        pl.groups_main = pl.ntiles / pl.rt_main;
        pl.rt_rem = pl.ntiles % pl.rt_main;
    }
    return pl;
}

template <int RT, int NOFF, bool EVENK>
hipError_t launch_fused_inst(const FusedParams& prm, int groups, size_t lds, int grid_x, hipStream_t st) {
    auto kfn = prm.dY ? leaf_fused_kernel<RT, NOFF, EVENK, true> : leaf_fused_kernel<RT, NOFF, EVENK, false>;
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kfn), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(kfn, dim3(grid_x, groups), dim3(kWavesPerWG * 64), lds, st, prm);
    return hipGetLastError();
}

template <int RT>
hipError_t launch_fused_rt(const FusedParams& prm, int noff_t, int groups, size_t lds, int grid_x, hipStream_t st) {
    const bool even = (prm.K % 2) == 0;
    switch (noff_t) {
        case 1: return even ? launch_fused_inst<RT, 1, true>(prm, groups, lds, grid_x, st)
                            : launch_fused_inst<RT, 1, false>(prm, groups, lds, grid_x, st);
        case 3: return even ? launch_fused_inst<RT, 3, true>(prm, groups, lds, grid_x, st)
                            : launch_fused_inst<RT, 3, false>(prm, groups, lds, grid_x, st);
        case 6: return even ? launch_fused_inst<RT, 6, true>(prm, groups, lds, grid_x, st)
                            : launch_fused_inst<RT, 6, false>(prm, groups, lds, grid_x, st);
    }
    return hipErrorInvalidValue;
}

hipError_t launch_fused(const FusedParams& prm, int rt, int noff_t, int groups, size_t lds, int grid_x, hipStream_t st) {
    switch (rt) {
        case 1: return launch_fused_rt<1>(prm, noff_t, groups, lds, grid_x, st);
        case 2: return launch_fused_rt<2>(prm, noff_t, groups, lds, grid_x, st);
        case 3: return launch_fused_rt<3>(prm, noff_t, groups, lds, grid_x, st);
    }
    return hipErrorInvalidValue;
}

struct BwdPlan {
    bool ok;
    int NKT, NW, TPW, NS, HPc, XSC, LD, nch;
    size_t lds;
};

BwdPlan make_bwd_plan(const FusedPlan& pl, int T) {
    BwdPlan bp{};
    if (!pl.ok) return bp;
    bp.NKT = ceil_div(pl.R, 16);
    bp.NW = 16;                                        // 4 waves per SIMD; k-tiles are dealt to them SIMD-balanced in-kernel
    bp.TPW = ceil_div(bp.NKT, bp.NW);
    bp.HPc = 16 * bp.NKT;
    bp.LD = 2 * pl.FP + 16;
    bp.NS = 0;
    for (int ns = 64; ns >= 16; ns >>= 1)
        if (ns * (2 * pl.FP) / 4 <= kDtPF * bp.NW * 64) { bp.NS = ns; break; }
    if (bp.NS == 0 || bp.TPW > 3) return bp;
    bp.XSC = bp.NS + 2 * bp.HPc;
    if (bp.XSC > 2 * bp.NW * 64) return bp;
    bp.lds = (size_t)2 * ((bp.XSC + 3) / 4 * 4 + (size_t)bp.NS * bp.LD) * 4;
    if (bp.lds > (size_t)kMaxLds) return bp;
    bp.nch = ceil_div(T, bp.NS);
    bp.ok = true;
    return bp;
}

template <int RT, int TPW>
hipError_t launch_dtaps_inst(const DtapsParams& prm, int groups, size_t lds, int grid_x, hipStream_t st) {
    auto kfn = (prm.K % 2) == 0 ? dtaps_mfma_kernel<RT, TPW, true> : dtaps_mfma_kernel<RT, TPW, false>;
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kfn), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(kfn, dim3(grid_x, groups), dim3(prm.NW * 64), lds, st, prm);
    return hipGetLastError();
}

template <int RT>
hipError_t launch_dtaps_rt(const DtapsParams& prm, int tpw, int groups, size_t lds, int grid_x, hipStream_t st) {
    switch (tpw) {
        case 1: return launch_dtaps_inst<RT, 1>(prm, groups, lds, grid_x, st);
        case 2: return launch_dtaps_inst<RT, 2>(prm, groups, lds, grid_x, st);
        case 3: return launch_dtaps_inst<RT, 3>(prm, groups, lds, grid_x, st);
    }
    return hipErrorInvalidValue;
}

hipError_t launch_dtaps(const DtapsParams& prm, int rt, int tpw, int groups, size_t lds, int grid_x, hipStream_t st) {
    switch (rt) {
        case 1: return launch_dtaps_rt<1>(prm, tpw, groups, lds, grid_x, st);
        case 2: return launch_dtaps_rt<2>(prm, tpw, groups, lds, grid_x, st);
        case 3: return launch_dtaps_rt<3>(prm, tpw, groups, lds, grid_x, st);
    }
    return hipErrorInvalidValue;
}

// the fused kernel over all filter groups: full groups of rt_main tiles, then the remainder group
hipError_t launch_fused_groups(FusedParams prm, const FusedPlan& pl, hipStream_t st) {
    const int cus = num_cus();
    const int wg_needed = ceil_div(prm.total_tasks, kWavesPerWG);
    if (pl.groups_main > 0) {
        prm.tile_base = 0;
        const int gx = std::max(1, std::min(wg_needed, std::max(1, cus / pl.groups_main)));
        hipError_t e = launch_fused(prm, pl.rt_main, pl.noff_t, pl.groups_main, fused_lds_bytes(pl.R, pl.rt_main, pl.XS), gx, st);
        if (e != hipSuccess) return e;
    }
    if (pl.rt_rem > 0) {
        prm.tile_base = pl.groups_main * pl.rt_main;
        const int gx = std::max(1, std::min(wg_needed, cus));
        hipError_t e = launch_fused(prm, pl.rt_rem, pl.noff_t, 1, fused_lds_bytes(pl.R, pl.rt_rem, pl.XS), gx, st);
        if (e != hipSuccess) return e;
    }
    return hipSuccess;
}

FusedParams base_fused_params(const FusedPlan& pl, int B, int T, int F, int K, int hop, int tuning_desync) {
    FusedParams prm{};
    prm.B = B; prm.T = T; prm.TP = pl.TP; prm.F = F; prm.FP = pl.FP; prm.K = K; prm.hop = hop; prm.padL = pl.padL;
    prm.KS = pl.KS; prm.Hf = pl.Hf; prm.xshift = pl.xshift; prm.NU = pl.NU; prm.HP = pl.HP; prm.XS = pl.XS;
    prm.q_lo = pl.q_lo; prm.nq = pl.nq; prm.noff = pl.noff; prm.total_tasks = B * pl.nq; prm.GJ = pl.GJ;
    // half a unit of MFMA work is ~ 16*kUB samples x 2*RT tiles x KS k-steps x 32 cycles; s_sleep(127) ~ 8.1k cycles
    prm.desync_sleeps = tuning_desync >= 0 ? tuning_desync
                                           : std::max(1, (int)((long long)kUB * 2 * pl.rt_main * pl.KS * 32 / 2 / 8128));
    return prm;
}

// workspace layout of the fused backward (floats)
struct BwdLayout {
    size_t W, G, Gs, meta, part, raw, ema, gpre, gcols, rowsum, dwpart, dHpart, dY, total;
};

BwdLayout bwd_layout(const FusedPlan& pl, const BwdPlan& bp, int B, int T, int F, int cus) {
    BwdLayout L{};
    size_t o = 0;
    auto take = [&](size_t n) { const size_t at = o; o += align_up(n, 64); return at; };
    L.W = take(pl.w_floats);
    L.G = take(pl.g_floats);
    L.Gs = take(pl.g_floats);
    L.meta = take(pl.meta_ints);
    L.part = take(pl.part_floats);
    L.raw = take((size_t)B * F * pl.TP);
    L.ema = take((size_t)B * F * pl.TP);
    L.gpre = take((size_t)B * F * pl.TP);
    L.gcols = take((size_t)B * pl.TP * pl.FP);
    L.rowsum = take((size_t)B * F * 4);
    L.dwpart = take((size_t)cus * kWavesPerWG * pl.FP);
    L.dHpart = take((size_t)(cus + 1) * 16 * bp.NKT * 2 * pl.FP);     // + 1 slab for the reduced sum
    L.dY = take((size_t)B * T * 2 * pl.FP);
    L.total = o;
    return L;
}

inline bool misaligned(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 3u) != 0; }

int check_shape(int B, int T, int F, int K, int hop) {
    if (B < 1 || T < 1 || F < 1 || K < 1 || hop < 1) return LEAF_ERR_BAD_SHAPE;
    if ((long long)B * T >= (1ll << 31)) return LEAF_ERR_BAD_SHAPE;
    return LEAF_OK;
}

size_t staged_workspace_floats(int B, int T, int F, int K, int hop) {
    const int padL = K / 2 + K % 2 - 1;
    const int TP = (T + (K - 1) - K) / hop + 1;
    (void)padL;
    return align_up((size_t)2 * F * K, 64) + align_up((size_t)F * K, 64) + align_up((size_t)B * 2 * F * T, 64) +
           align_up((size_t)B * F * T, 64) + align_up((size_t)B * F * TP, 64);
}

#define LEAF_LAUNCH_CHECK()                                  \
    do {                                                     \
        if (hipGetLastError() != hipSuccess) return LEAF_ERR_LAUNCH; \
    } while (0)

}  // namespace

extern "C" {

int leaf_abi_version(void) { return LEAF_ABI_VERSION; }

const char* leaf_status_string(int status) {
    switch (status) {
        case LEAF_OK: return "ok";
        case LEAF_ERR_NULL_POINTER: return "null pointer argument";
        case LEAF_ERR_BAD_SHAPE: return "bad shape (B,T,F,K,hop must be >= 1 and B*T < 2^31)";
        case LEAF_ERR_WORKSPACE: return "workspace missing or too small (see leaf_workspace_bytes)";
        case LEAF_ERR_BAD_ALGO: return "unknown or inapplicable algorithm selector";
        case LEAF_ERR_LAUNCH: return "HIP kernel launch failed";
        case LEAF_ERR_NO_DEVICE: return "no usable gfx950 device";
        case LEAF_ERR_ALIGNMENT: return "buffer not 4-byte aligned";
    }
    return "unknown status";
}

int leaf_num_frames(int T, int K, int hop) {
    if (T < 1 || K < 1 || hop < 1) return LEAF_ERR_BAD_SHAPE;
    const int padL = K / 2 + K % 2 - 1, padR = K / 2;
    return (T + padL + padR - K) / hop + 1;
}

size_t leaf_workspace_bytes(int B, int T, int F, int K, int hop, int algo) {
    if (check_shape(B, T, F, K, hop) != LEAF_OK) return 0;
    const FusedPlan pl = make_plan(B, T, F, K, hop);
    const size_t fused = pl.ok ? (align_up(pl.w_floats, 64) + align_up(pl.g_floats, 64) + align_up(pl.meta_ints, 64) +
                                  align_up(pl.part_floats, 64) + (LEAF_TRACE ? 8 * 64 * 2 : 0)) * 4
                               : 0;
    const size_t staged = staged_workspace_floats(B, T, F, K, hop) * 4;
    if (algo == LEAF_ALGO_MFMA) return fused;
    if (algo == LEAF_ALGO_STAGED) return staged;
    if (algo == LEAF_ALGO_AUTO) return pl.ok ? fused : staged;
    return 0;
}

int leaf_gabor_taps_f32(const float* kernel, int F, int K, float* taps, void* stream) {
    if (!kernel || !taps) return LEAF_ERR_NULL_POINTER;
    if (F < 1 || K < 1) return LEAF_ERR_BAD_SHAPE;
    hipLaunchKernelGGL(taps_direct_kernel, dim3(ceil_div(F * K, 256)), dim3(256), 0, (hipStream_t)stream, kernel, F, K,
                       gabor_bounds(K), taps);
    LEAF_LAUNCH_CHECK();
    return LEAF_OK;
}

int leaf_lowpass_window_f32(const float* pool_w, int F, int K, float* window, void* stream) {
    if (!pool_w || !window) return LEAF_ERR_NULL_POINTER;
    if (F < 1 || K < 1) return LEAF_ERR_BAD_SHAPE;
    hipLaunchKernelGGL(lowpass_window_kernel, dim3(ceil_div(F * K, 256)), dim3(256), 0, (hipStream_t)stream, pool_w, F, K,
                       window);
    LEAF_LAUNCH_CHECK();
    return LEAF_OK;
}

int leaf_gabor_conv_f32(const float* x, int B, int T, const float* kernel, int F, int K, float* y, void* workspace,
                        size_t workspace_bytes, void* stream) {
    if (!x || !kernel || !y) return LEAF_ERR_NULL_POINTER;
    if (check_shape(B, T, F, K, 1) != LEAF_OK || 2 * F > 65535 || B > 65535) return LEAF_ERR_BAD_SHAPE;
    if (!workspace || workspace_bytes < (size_t)2 * F * K * 4) return LEAF_ERR_WORKSPACE;
    float* taps = static_cast<float*>(workspace);
    int rc = leaf_gabor_taps_f32(kernel, F, K, taps, stream);
    if (rc != LEAF_OK) return rc;
    const int padL = K / 2 + K % 2 - 1;
    hipLaunchKernelGGL(conv_staged_kernel, dim3(ceil_div(T, 256), 2 * F, B), dim3(256), 0, (hipStream_t)stream, x, taps, B,
                       T, 2 * F, K, padL, y);
    LEAF_LAUNCH_CHECK();
    return LEAF_OK;
}

int leaf_squared_modulus_f32(const float* y, int B, int F, int T, float* e, void* stream) {
    if (!y || !e) return LEAF_ERR_NULL_POINTER;
    if (B < 1 || F < 1 || T < 1) return LEAF_ERR_BAD_SHAPE;
    const size_t n = (size_t)B * F * T;
    hipLaunchKernelGGL(sqmod_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, y,
                       (size_t)B * F, T, e);
    LEAF_LAUNCH_CHECK();
    return LEAF_OK;
}

int leaf_gaussian_lowpass_f32(const float* e, int B, int F, int T, const float* pool_w, const float* pool_b, int K,
                              int hop, float* pooled, void* workspace, size_t workspace_bytes, void* stream) {
    if (!e || !pool_w || !pooled) return LEAF_ERR_NULL_POINTER;
    if (check_shape(B, T, F, K, hop) != LEAF_OK || F > 65535 || B > 65535) return LEAF_ERR_BAD_SHAPE;
    if (!workspace || workspace_bytes < (size_t)F * K * 4) return LEAF_ERR_WORKSPACE;
    float* g = static_cast<float*>(workspace);
    int rc = leaf_lowpass_window_f32(pool_w, F, K, g, stream);
    if (rc != LEAF_OK) return rc;
    const int TP = leaf_num_frames(T, K, hop);
    const int padL = K / 2 + K % 2 - 1;
    hipLaunchKernelGGL(pool_staged_kernel, dim3(ceil_div(TP, 64), F, B), dim3(64), 0, (hipStream_t)stream, e, g, pool_b, F,
                       T, TP, K, hop, padL, pooled);
    LEAF_LAUNCH_CHECK();
    return LEAF_OK;
}

int leaf_ema_f32(const float* p, int B, int F, int TP, const float* ema_w, float* ema, void* stream) {
    if (!p || !ema_w || !ema) return LEAF_ERR_NULL_POINTER;
    if (B < 1 || F < 1 || TP < 1) return LEAF_ERR_BAD_SHAPE;
    hipLaunchKernelGGL(pcen_rows_kernel, dim3(ceil_div(B * F, 64)), dim3(64), 0, (hipStream_t)stream, p, B * F, F, TP,
                       (const float*)nullptr, (const float*)nullptr, (const float*)nullptr, ema_w, 0.0f, 0, ema);
    LEAF_LAUNCH_CHECK();
    return LEAF_OK;
}

int leaf_pcen_f32(const float* p, int B, int F, int TP, const float* alpha, const float* delta, const float* root,
                  const float* ema_w, float floor_, float* out, void* stream) {
    if (!p || !alpha || !delta || !root || !ema_w || !out) return LEAF_ERR_NULL_POINTER;
    if (B < 1 || F < 1 || TP < 1) return LEAF_ERR_BAD_SHAPE;
    hipLaunchKernelGGL(pcen_rows_kernel, dim3(ceil_div(B * F, 64)), dim3(64), 0, (hipStream_t)stream, p, B * F, F, TP,
                       alpha, delta, root, ema_w, floor_, 1, out);
    LEAF_LAUNCH_CHECK();
    return LEAF_OK;
}

static int forward_impl(const void* x, int B, int T, const float* kernel, const float* pool_w, const float* pool_b,
                        const float* alpha, const float* delta, const float* root, const float* ema_w, int F, int K, int hop,
                        int flags, int algo, void* out, void* workspace, size_t workspace_bytes, void* stream,
                        hipEvent_t* ev, float* pooled_raw = nullptr) {
    if (!x || !kernel || !pool_w || !pool_b || !out) return LEAF_ERR_NULL_POINTER;
    const bool use_pcen = (flags & LEAF_FLAG_PCEN) != 0;
    if (use_pcen && (!alpha || !delta || !root || !ema_w)) return LEAF_ERR_NULL_POINTER;
    int rc = check_shape(B, T, F, K, hop);
    if (rc != LEAF_OK) return rc;
    {
        const uintptr_t io_mask = (flags & LEAF_FLAG_IO_BF16) ? 1u : 3u;
        if ((reinterpret_cast<uintptr_t>(x) & io_mask) || (reinterpret_cast<uintptr_t>(out) & io_mask) || misaligned(workspace))
            return LEAF_ERR_ALIGNMENT;
    }
    const int tuning_desync = ((algo >> 8) & 0xff) - 1;      // LEAF_ALGO_TUNE_DESYNC(n); -1 = automatic
    algo &= 0xff;
    if (algo != LEAF_ALGO_AUTO && algo != LEAF_ALGO_STAGED && algo != LEAF_ALGO_MFMA) return LEAF_ERR_BAD_ALGO;
    const FusedPlan pl = make_plan(B, T, F, K, hop);
    const bool io_bf16 = (flags & LEAF_FLAG_IO_BF16) != 0;
    if (io_bf16 && (algo == LEAF_ALGO_STAGED || !pl.ok)) return LEAF_ERR_BAD_ALGO;   // bf16 I/O is a fused-path feature
    if (algo == LEAF_ALGO_MFMA && !pl.ok) return LEAF_ERR_BAD_ALGO;
    if (algo == LEAF_ALGO_AUTO) algo = pl.ok ? LEAF_ALGO_MFMA : LEAF_ALGO_STAGED;
    const size_t need = leaf_workspace_bytes(B, T, F, K, hop, algo);
    if (!workspace || workspace_bytes < need) return LEAF_ERR_WORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    const int mode = (use_pcen ? 1 : 0) | ((flags & LEAF_FLAG_LOG1P) && !use_pcen ? 2 : 0) | (io_bf16 ? 4 : 0);
    const int TP = pl.TP;
    float* ws = static_cast<float*>(workspace);

    if (algo == LEAF_ALGO_MFMA) {
        float* W = ws;
        float* G = ws + align_up(pl.w_floats, 64);
        int* meta = reinterpret_cast<int*>(G + align_up(pl.g_floats, 64));
        int* perm = meta;
        int* col_of = meta + pl.FP;
        int* tile_ks = meta + 2 * pl.FP;
        float* part = G + align_up(pl.g_floats, 64) + align_up(pl.meta_ints, 64);
        if (ev) (void)hipEventRecord(ev[0], st);
        hipLaunchKernelGGL(fused_prep_kernel, dim3(ceil_div(pl.R * 2 * pl.FP + pl.FP * pl.GJ, 256)), dim3(256), 0, st,
                           kernel, pool_w, F, pl.FP, K, pl.R, pl.GJ, gabor_bounds(K), W, G, (float*)nullptr, perm, col_of,
                           tile_ks);
        LEAF_LAUNCH_CHECK();
        if (ev) (void)hipEventRecord(ev[1], st);
        FusedParams prm = base_fused_params(pl, B, T, F, K, hop, tuning_desync);
        prm.x = x; prm.io_bf16 = io_bf16 ? 1 : 0; prm.W = W; prm.G = G; prm.tile_ks = tile_ks; prm.part = part;
#if LEAF_TRACE
        prm.trace = reinterpret_cast<unsigned long long*>(part + align_up(pl.part_floats, 64));
#endif
        if (launch_fused_groups(prm, pl, st) != hipSuccess) return LEAF_ERR_LAUNCH;
        if (ev) (void)hipEventRecord(ev[2], st);
        hipLaunchKernelGGL(finalize_kernel, dim3(B), dim3(kFinThreads), (size_t)F * 68 * 4, st, part, F, pl.FP, TP, pl.noff,
                           pl.q_lo, pl.q_hi, col_of, pool_b, alpha, delta, root, ema_w, 1e-12f, mode, out, pooled_raw);
        LEAF_LAUNCH_CHECK();
        if (ev) (void)hipEventRecord(ev[3], st);
        return LEAF_OK;
    }

    // staged path: every intermediate of the reference graph is materialised in the workspace
    if (2 * F > 65535 || B > 65535) return LEAF_ERR_BAD_SHAPE;
    const float* xf32 = static_cast<const float*>(x);
    float* outf32 = static_cast<float*>(out);
    float* taps = ws;
    float* g = taps + align_up((size_t)2 * F * K, 64);
    float* y = g + align_up((size_t)F * K, 64);
    float* e = y + align_up((size_t)B * 2 * F * T, 64);
    float* pooled = e + align_up((size_t)B * F * T, 64);
    rc = leaf_gabor_conv_f32(xf32, B, T, kernel, F, K, y, taps, (size_t)2 * F * K * 4, stream);
    if (rc != LEAF_OK) return rc;
    rc = leaf_squared_modulus_f32(y, B, F, T, e, stream);
    if (rc != LEAF_OK) return rc;
    rc = leaf_gaussian_lowpass_f32(e, B, F, T, pool_w, pool_b, K, hop, pooled, g, (size_t)F * K * 4, stream);
    if (rc != LEAF_OK) return rc;
    const size_t n = (size_t)B * F * TP;
    if (pooled_raw && hipMemcpyAsync(pooled_raw, pooled, n * 4, hipMemcpyDeviceToDevice, st) != hipSuccess)
        return LEAF_ERR_LAUNCH;
    if (use_pcen) {
        // floor in place, then PCEN
        hipLaunchKernelGGL(floor_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, pooled, n, 0, pooled);
        LEAF_LAUNCH_CHECK();
        return leaf_pcen_f32(pooled, B, F, TP, alpha, delta, root, ema_w, 1e-12f, outf32, stream);
    }
    hipLaunchKernelGGL(floor_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, pooled, n, mode, outf32);
    LEAF_LAUNCH_CHECK();
    return LEAF_OK;
}

int leaf_forward_f32(const float* x, int B, int T, const float* kernel, const float* pool_w, const float* pool_b,
                     const float* alpha, const float* delta, const float* root, const float* ema_w, int F, int K, int hop,
                     int flags, int algo, float* out, void* workspace, size_t workspace_bytes, void* stream) {
    return forward_impl(x, B, T, kernel, pool_w, pool_b, alpha, delta, root, ema_w, F, K, hop, flags, algo, out, workspace,
                        workspace_bytes, stream, nullptr);
}

int leaf_forward_save_f32(const float* x, int B, int T, const float* kernel, const float* pool_w, const float* pool_b,
                          const float* alpha, const float* delta, const float* root, const float* ema_w, int F, int K, int hop,
                          int flags, int algo, float* out, float* pooled_raw, void* workspace, size_t workspace_bytes,
                          void* stream) {
    if (!pooled_raw) return LEAF_ERR_NULL_POINTER;
    if (flags & LEAF_FLAG_IO_BF16) return LEAF_ERR_BAD_ALGO;
    return forward_impl(x, B, T, kernel, pool_w, pool_b, alpha, delta, root, ema_w, F, K, hop, flags, algo, out, workspace,
                        workspace_bytes, stream, nullptr, pooled_raw);
}

int leaf_forward_profiled_f32(const float* x, int B, int T, const float* kernel, const float* pool_w, const float* pool_b,
                              const float* alpha, const float* delta, const float* root, const float* ema_w, int F, int K,
                              int hop, int flags, float* out, void* workspace, size_t workspace_bytes, void* stream,
                              float* stage_ms) {
    if (!stage_ms) return LEAF_ERR_NULL_POINTER;
    hipEvent_t ev[4];
    for (int i = 0; i < 4; ++i)
        if (hipEventCreate(&ev[i]) != hipSuccess) return LEAF_ERR_LAUNCH;
    int rc = forward_impl(x, B, T, kernel, pool_w, pool_b, alpha, delta, root, ema_w, F, K, hop, flags, LEAF_ALGO_MFMA, out,
                          workspace, workspace_bytes, stream, ev);
    if (rc == LEAF_OK) {
        if (hipEventSynchronize(ev[3]) != hipSuccess) rc = LEAF_ERR_LAUNCH;
        for (int i = 0; i < 3 && rc == LEAF_OK; ++i)
            if (hipEventElapsedTime(&stage_ms[i], ev[i], ev[i + 1]) != hipSuccess) rc = LEAF_ERR_LAUNCH;
    }
    for (int i = 0; i < 4; ++i) (void)hipEventDestroy(ev[i]);
    return rc;
}

size_t leaf_backward_workspace_bytes(int B, int T, int F, int K, int hop) {
    if (check_shape(B, T, F, K, hop) != LEAF_OK) return 0;
    const int TP = (T - 1) / hop + 1;
    const size_t fl = align_up((size_t)2 * F * K, 64) * 2 /* taps, dtaps unused slot */ + align_up((size_t)F * K, 64) * 2 +
                      align_up((size_t)B * 2 * F * T, 64) + align_up((size_t)B * F * T, 64) +
                      align_up((size_t)B * F * TP, 64) * 3 + align_up((size_t)B * F * 4, 64) +
                      align_up((size_t)B * 2 * F * K, 64);
    const FusedPlan pl = make_plan(B, T, F, K, hop);
    const BwdPlan bp = make_bwd_plan(pl, T);
    size_t fused = 0;
    if (bp.ok) fused = bwd_layout(pl, bp, B, T, F, num_cus()).total;
    return std::max(fl, fused) * 4;
}

int leaf_backward_f32(const float* x, int B, int T, const float* kernel, const float* pool_w, const float* pool_b,
                      const float* alpha, const float* delta, const float* root, const float* ema_w, int F, int K, int hop,
                      int flags, const float* grad_out, const float* pooled_raw, float* g_kernel, float* g_pool_w,
                      float* g_pool_b, float* g_alpha, float* g_delta, float* g_root, float* g_ema_w, float* g_x,
                      void* workspace, size_t workspace_bytes, void* stream) {
    if (!x || !kernel || !pool_w || !pool_b || !grad_out || !g_kernel || !g_pool_w || !g_pool_b) return LEAF_ERR_NULL_POINTER;
    const bool use_pcen = (flags & LEAF_FLAG_PCEN) != 0;
    if (use_pcen && (!alpha || !delta || !root || !ema_w || !g_alpha || !g_delta || !g_root || !g_ema_w))
        return LEAF_ERR_NULL_POINTER;
    int rc = check_shape(B, T, F, K, hop);
    if (rc != LEAF_OK) return rc;
    if (2 * F > 65535 || B > 65535) return LEAF_ERR_BAD_SHAPE;
    if (!workspace || workspace_bytes < leaf_backward_workspace_bytes(B, T, F, K, hop)) return LEAF_ERR_WORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    const int TP = (T - 1) / hop + 1;
    const int padL = K / 2 + K % 2 - 1;
    const int mode = use_pcen ? 1 : 0;
    float* ws = static_cast<float*>(workspace);
    {
        // ---- fused backward (MFMA): used whenever the geometry fits and dL/dx is not requested
        const FusedPlan pl = make_plan(B, T, F, K, hop);
        const BwdPlan bp = make_bwd_plan(pl, T);
        if (bp.ok && !g_x && !(flags & LEAF_FLAG_BWD_STAGED)) {
            const int cus = num_cus();
            const BwdLayout L = bwd_layout(pl, bp, B, T, F, cus);
            float* W = ws + L.W; float* G = ws + L.G; float* Gs = ws + L.Gs;
            int* meta = reinterpret_cast<int*>(ws + L.meta);
            int* perm = meta; int* col_of = meta + pl.FP; int* tile_ks = meta + 2 * pl.FP;
            float* part = ws + L.part; float* raw = ws + L.raw; float* ema = ws + L.ema; float* gpre = ws + L.gpre;
            float* gcols = ws + L.gcols; float* rowsum = ws + L.rowsum; float* dwpart = ws + L.dwpart;
            float* dHpart = ws + L.dHpart; float* dY = ws + L.dY;
            const int Rp = 16 * bp.NKT;
            // 1. tables, forward recompute up to the pre-floor pooled value
            hipLaunchKernelGGL(fused_prep_kernel, dim3(ceil_div(pl.R * 2 * pl.FP + pl.FP * pl.GJ, 256)), dim3(256), 0, st,
                               kernel, pool_w, F, pl.FP, K, pl.R, pl.GJ, gabor_bounds(K), W, G, Gs, perm, col_of, tile_ks);
            LEAF_LAUNCH_CHECK();
            FusedParams prm = base_fused_params(pl, B, T, F, K, hop, -1);
            prm.x = x; prm.W = W; prm.G = G; prm.tile_ks = tile_ks; prm.part = part;
            const float* raw_in = pooled_raw;          // saved by leaf_forward_save_f32, else recomputed here
            if (!raw_in) {
                if (launch_fused_groups(prm, pl, st) != hipSuccess) return LEAF_ERR_LAUNCH;
                hipLaunchKernelGGL(finalize_kernel, dim3(B), dim3(kFinThreads), (size_t)F * 68 * 4, st, part, F, pl.FP, TP,
                                   pl.noff, pl.q_lo, pl.q_hi, col_of, pool_b, alpha, delta, root, ema_w, 1e-12f, 8, raw,
                                   (float*)nullptr);
                LEAF_LAUNCH_CHECK();
                raw_in = raw;
            }
            // 2. floor + PCEN backward per (b,f) row
            if (hipMemsetAsync(gcols, 0, (size_t)B * TP * pl.FP * 4, st) != hipSuccess) return LEAF_ERR_LAUNCH;
            hipLaunchKernelGGL(pcen_bwd_rows_kernel, dim3(ceil_div(B * F, 64)), dim3(64), 0, st, raw_in, grad_out, B * F, F, TP,
                               alpha, delta, root, ema_w, 1e-12f, mode, ema, gpre, rowsum, col_of, pl.FP, gcols);
            LEAF_LAUNCH_CHECK();
            // 3. filterbank recompute with the backward epilogue: dY (time-major) and d pool_w partials
            if (hipMemsetAsync(dwpart, 0, (size_t)cus * kWavesPerWG * pl.FP * 4, st) != hipSuccess) return LEAF_ERR_LAUNCH;
            if (hipMemsetAsync(dHpart, 0, (size_t)cus * Rp * 2 * pl.FP * 4, st) != hipSuccess) return LEAF_ERR_LAUNCH;
            prm.Gs = Gs; prm.gcols = gcols; prm.dY = dY; prm.dwpart = dwpart; prm.part = nullptr;
            if (launch_fused_groups(prm, pl, st) != hipSuccess) return LEAF_ERR_LAUNCH;
            // 4. tap gradients: dH = S^T dY on the MFMA, then chain to (mu, sigma)
            DtapsParams dp{};
            dp.x = x; dp.dY = dY; dp.tile_ks = tile_ks; dp.dHpart = dHpart;
            dp.B = B; dp.T = T; dp.FP = pl.FP; dp.K = K; dp.Hf = pl.Hf; dp.xshift = pl.xshift;
            dp.NKT = bp.NKT; dp.NW = bp.NW; dp.NS = bp.NS; dp.HPc = bp.HPc; dp.XSC = bp.XSC; dp.LD = bp.LD;
            dp.nch = bp.nch; dp.total_chunks = B * bp.nch;
            if (pl.groups_main > 0) {
                dp.tile_base = 0;
                const int gx = std::max(1, std::min(dp.total_chunks, std::max(1, cus / pl.groups_main)));
                if (launch_dtaps(dp, pl.rt_main, bp.TPW, pl.groups_main, bp.lds, gx, st) != hipSuccess) return LEAF_ERR_LAUNCH;
            }
            if (pl.rt_rem > 0) {
                dp.tile_base = pl.groups_main * pl.rt_main;
                const int gx = std::max(1, std::min(dp.total_chunks, cus));
                if (launch_dtaps(dp, pl.rt_rem, bp.TPW, 1, bp.lds, gx, st) != hipSuccess) return LEAF_ERR_LAUNCH;
            }
            {
                // slab 0 doubles as the reduction target: reduce slabs 1.. into a scratch slab placed after the last one
                const size_t slab = (size_t)Rp * 2 * pl.FP;
                float* dHsum = dHpart + (size_t)cus * slab;
                hipLaunchKernelGGL(dh_reduce_kernel, dim3((unsigned)((slab + 255) / 256)), dim3(256), 0, st, dHpart, cus, slab,
                                   dHsum);
                LEAF_LAUNCH_CHECK();
                hipLaunchKernelGGL(dkernel_fused_kernel, dim3(F), dim3(256), 0, st, dHsum, 1, Rp, W, pl.R, pl.FP, col_of,
                                   kernel, F, gabor_bounds(K), g_kernel);
            }
            LEAF_LAUNCH_CHECK();
            // 5. parameter sums over the batch
            hipLaunchKernelGGL(param_reduce_kernel, dim3(F), dim3(256), 0, st, gpre, (const float*)nullptr,
                               (const float*)nullptr, rowsum, pool_w, B, F, TP, K, mode, dwpart, cus * kWavesPerWG, pl.FP,
                               col_of, g_pool_w, g_pool_b, g_alpha, g_delta, g_root, g_ema_w);
            LEAF_LAUNCH_CHECK();
            return LEAF_OK;
        }
    }
    float* taps = ws;                         ws += align_up((size_t)2 * F * K, 64) * 2;
    float* g = ws;                            ws += align_up((size_t)F * K, 64);
    float* dg = ws;                           ws += align_up((size_t)F * K, 64);
    float* y = ws;                            ws += align_up((size_t)B * 2 * F * T, 64);
    float* e = ws;                            ws += align_up((size_t)B * F * T, 64);
    float* raw = ws;                          ws += align_up((size_t)B * F * TP, 64);
    float* ema = ws;                          ws += align_up((size_t)B * F * TP, 64);
    float* gpre = ws;                         ws += align_up((size_t)B * F * TP, 64);
    float* rowsum = ws;                       ws += align_up((size_t)B * F * 4, 64);
    float* tpart = ws;
    // forward recompute (staged kernels = the reference graph)
    rc = leaf_gabor_conv_f32(x, B, T, kernel, F, K, y, taps, (size_t)2 * F * K * 4, stream);
    if (rc != LEAF_OK) return rc;
    rc = leaf_squared_modulus_f32(y, B, F, T, e, stream);
    if (rc != LEAF_OK) return rc;
    rc = leaf_gaussian_lowpass_f32(e, B, F, T, pool_w, pool_b, K, hop, raw, g, (size_t)F * K * 4, stream);
    if (rc != LEAF_OK) return rc;
    // PCEN + floor backward
    hipLaunchKernelGGL(pcen_bwd_rows_kernel, dim3(ceil_div(B * F, 64)), dim3(64), 0, st, raw, grad_out, B * F, F, TP, alpha,
                       delta, root, ema_w, 1e-12f, mode, ema, gpre, rowsum, (const int*)nullptr, 0, (float*)nullptr);
    LEAF_LAUNCH_CHECK();
    // pooling backward: window gradient needs e, sample gradient turns y into dy in place
    hipLaunchKernelGGL(pool_bwd_dg_kernel, dim3(ceil_div(K, 128), F), dim3(128), 0, st, e, gpre, B, F, T, TP, K, hop, padL, dg);
    LEAF_LAUNCH_CHECK();
    hipLaunchKernelGGL(param_reduce_kernel, dim3(F), dim3(256), 0, st, gpre, dg, g, rowsum, pool_w, B, F, TP, K, mode,
                       (const float*)nullptr, 0, 0, (const int*)nullptr, g_pool_w, g_pool_b, g_alpha, g_delta, g_root, g_ema_w);
    LEAF_LAUNCH_CHECK();
    hipLaunchKernelGGL(pool_bwd_dy_kernel, dim3(ceil_div(T, 256), F, B), dim3(256), 0, st, y, g, gpre, F, T, TP, K, hop, padL);
    LEAF_LAUNCH_CHECK();
    // filterbank backward
    hipLaunchKernelGGL(dtaps_partial_kernel, dim3(ceil_div(K, 128), 2 * F, B), dim3(128), 0, st, y, x, T, 2 * F, K, padL,
                       tpart);
    LEAF_LAUNCH_CHECK();
    hipLaunchKernelGGL(dkernel_kernel, dim3(F), dim3(256), 0, st, tpart, taps, kernel, B, F, K, gabor_bounds(K), g_kernel);
    LEAF_LAUNCH_CHECK();
    if (g_x) {
        hipLaunchKernelGGL(dx_kernel, dim3(ceil_div(T, 256), B), dim3(256), 0, st, y, taps, T, 2 * F, K, padL, g_x);
        LEAF_LAUNCH_CHECK();
    }
    return LEAF_OK;
}

}  // extern "C"
