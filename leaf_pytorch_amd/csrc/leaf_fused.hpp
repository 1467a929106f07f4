// leaf_fused.hpp -- fused forward path: prep tables, MFMA filterbank+pool kernel (also its backward epilogue), finalize/PCEN
// Part of the single translation unit leaf_kernels.hip (gfx950 only); see that file's header comment.
#pragma once
#include "leaf_common.hpp"

namespace {

// ---------------------------------------------------------------------------------------------
// fused path: prep (filter ordering + half-support tap table), fused filterbank/pool kernel, finalize
// ---------------------------------------------------------------------------------------------

// Taps smaller than exp(-kTapCut^2/2) = 1.5e-8 of a filter's peak are not issued: the Gaussian envelope
// puts them below the fp32 rounding noise of the 400-term sums they would join (NOTES.md section 2).
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
#ifndef LEAF_INST_TU               // non-template kernel: compiled once, in leaf_kernels.hip
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
#endif

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

// Layout fingerprints of the parameter structs that cross translation units as opaque kernel arguments (leaf_inst.hpp): the
// structs live in the headers' unnamed namespaces, so every inst_*.hip has its own copy of the type, and a macro that changed
// one copy only (LEAF_TRACE, LEAF_TOOLS, ...) would shift fields silently.  Each inst_*.hip exports the fingerprint it was
// compiled with (leaf_layout_*), leaf_kernels.hip compares them with its own before the first launch.
constexpr unsigned leaf_mix(unsigned h, size_t v) { return (h ^ (unsigned)v) * 16777619u; }
constexpr unsigned leaf_layout_hash_fused() {
    unsigned h = 2166136261u;
    h = leaf_mix(h, sizeof(FusedParams)); h = leaf_mix(h, offsetof(FusedParams, part)); h = leaf_mix(h, offsetof(FusedParams, trace));
    h = leaf_mix(h, offsetof(FusedParams, dwpart));
    return h;
}

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
// (postprocessing.py:13-28, 62-69).  One workgroup per (clip, group of kFinGroup filters), 128-frame chunks (a 1 s
// clip is one chunk):
//   phase 1  all threads: pooled[f][m] -> LDS (partial reads coalesced across filters)
//   phase 2  one wave per filter, lane l owns frames 2l and 2l+1: the first-order recurrence
//            M_m = w p_m + (1-w) M_{m-1} is an affine map composition -- composed in-lane for the pair, scanned
//            across the wavefront with 6 shuffle steps, with a carried state between chunks; PCEN is pointwise.
// mode bit0: PCEN, bit1: log1p (extension), bit2: bf16 output, bit3: raw (pre-floor) output, bit4: every slot valid
// Which of a frame's `noff` partial slots hold data: bit4 -> all; SlotGeom.L > 0 (overlap-save path: slot s = s-th
// block the frame's window meets) -> computed from the geometry, so the partial buffer needs no zero fill; otherwise
// (direct path) slot dd is valid when hop-block q = m + dd lies in [q_lo, q_hi].
struct SlotGeom {
    int L, padL, K, hop, T;
    int nslot;                   // overlap-save path: slots per frame in the partial buffer (2 or 3)
};
constexpr int kFinThreads = 1024;   // launch bound; launched with 64 threads per filter of the group
constexpr int kFinGroup = 8;        // filters per workgroup (one wave each)
constexpr int kFinPer = 4;        // pooled values a thread gathers per pass (independent loads in flight)
constexpr int kFinFrames = 128;   // frames per chunk (two per lane)
constexpr int kFinStride = kFinFrames + 1;
#ifndef LEAF_INST_TU               // non-template kernel: compiled once, in leaf_kernels.hip
__global__ __launch_bounds__(kFinThreads) void finalize_kernel(
    const float* __restrict__ part, int F, int FP, int TP, int noff, int q_lo, int q_hi, SlotGeom geo,
    const int* __restrict__ col_of, const float* __restrict__ bias, const float* __restrict__ alpha,
    const float* __restrict__ delta, const float* __restrict__ root, const float* __restrict__ ema_w, float floor_, int mode,
    void* __restrict__ out_, float* __restrict__ raw_out /* optional [B][F][TP]: bias + pooled sum before the floor */) {
    extern __shared__ __attribute__((aligned(16))) float fsm[];
    float* out = static_cast<float*>(out_);
    unsigned short* outh = static_cast<unsigned short*>(out_);
    // blockIdx.y = filter group: filters [f_lo, f_lo + nf), normally one per wave
    const int per = (F + (int)gridDim.y - 1) / (int)gridDim.y;
    const int f_lo = blockIdx.y * per, nf = min(F - f_lo, per);
    if (nf <= 0) return;
    float* sv = fsm;                         // [nf][kFinStride] pooled values of the current chunk
    float* scarry = sv + per * kFinStride;   // [nf] EMA state carried across chunks
    float* s_dr = scarry + per;              // [nf] delta^(1/r)
    int* s_col = reinterpret_cast<int*>(s_dr + per);   // [nf] tap column of each filter
    const int b = blockIdx.x, tid = threadIdx.x, nthreads = blockDim.x;
    const int wave = tid >> 6, lane = tid & 63;
    for (int fl = tid; fl < nf; fl += nthreads) {
        const int f = f_lo + fl;
        s_col[fl] = col_of[f];
        s_dr[fl] = (mode & 1) ? powf(delta[f], 1.0f / fmaxf(root[f], 1.0f)) : 0.0f;
    }
    __syncthreads();
    for (int m0 = 0; m0 < TP; m0 += kFinFrames) {
        const int nm = min(kFinFrames, TP - m0);
        for (int base = 0; base < nm * nf; base += nthreads * kFinPer) {
            float acc[kFinPer];
            int slot[kFinPer];
#pragma unroll
            for (int i = 0; i < kFinPer; ++i) {
                const int idx = base + i * nthreads + tid;
                acc[i] = 0.0f;
                slot[i] = -1;
                if (idx < nm * nf) {
                    const int mm = idx / nf, fl = idx - mm * nf, f = f_lo + fl;
                    const int m = m0 + mm;
                    const float* pp = part + (((size_t)b * TP + m) * noff) * FP + s_col[fl];
                    int nvalid = noff;
                    if (geo.L > 0) {
                        const int s0 = m * geo.hop - geo.padL;
                        nvalid = min(geo.T - 1, s0 + geo.K - 1) / geo.L - max(0, s0) / geo.L + 1;
                    }
                    for (int dd = 0; dd < noff; ++dd) {
                        const int q = m + dd;
                        const bool valid = geo.L > 0 ? dd < nvalid : ((mode & 16) || (q >= q_lo && q <= q_hi));
                        if (valid) acc[i] += pp[(size_t)dd * FP];
                    }
                    acc[i] += bias ? bias[f] : 0.0f;
                    slot[i] = fl * kFinStride + mm;
                }
            }
#pragma unroll
            for (int i = 0; i < kFinPer; ++i)
                if (slot[i] >= 0) {
                    sv[slot[i]] = (mode & 8) ? acc[i] : pooled_floor(acc[i]);
                    if (raw_out) {
                        const int fl = slot[i] / kFinStride, mm = slot[i] - fl * kFinStride;
                        raw_out[((size_t)b * F + f_lo + fl) * TP + m0 + mm] = acc[i];
                    }
                }
        }
        __syncthreads();
        for (int fl = wave; fl < nf; fl += nthreads / 64) {
            const int f = f_lo + fl;
            const int j0 = 2 * lane, j1 = j0 + 1;
            const float v0 = j0 < nm ? sv[fl * kFinStride + j0] : 0.0f;
            const float v1 = j1 < nm ? sv[fl * kFinStride + j1] : 0.0f;
            float r0 = v0, r1 = v1;
            if (mode & 8) {                              // backward: pre-floor pooled value
            } else if (mode & 1) {
                const float w = fminf(fmaxf(ema_w[f], 0.0f), 1.0f);
                const float A0 = j0 < nm ? 1.0f - w : 1.0f, B0 = j0 < nm ? w * v0 : 0.0f;   // M_m = A_m M_{m-1} + B_m
                const float A1 = j1 < nm ? 1.0f - w : 1.0f, B1 = j1 < nm ? w * v1 : 0.0f;
                float A = A1 * A0, Bv = fmaf(A1, B0, B1);    // the pair's map, then the inclusive scan over lanes
#pragma unroll
                for (int off = 1; off < 64; off <<= 1) {
                    const float Ap = __shfl_up(A, off), Bp = __shfl_up(Bv, off);
                    if (lane >= off) {
                        Bv = fmaf(A, Bp, Bv);
                        A *= Ap;
                    }
                }
                const float carry = (m0 == 0) ? sv[fl * kFinStride] : scarry[fl];   // state starts at p_0 (postprocessing.py:15)
                const float Mend = fmaf(A, carry, Bv);       // state after this lane's second frame
                const float Mprev = __shfl_up(Mend, 1);
                const float M0 = fmaf(A0, lane ? Mprev : carry, B0);
                const float M1 = fmaf(A1, M0, B1);
                const float last = __shfl(((nm - 1) & 1) ? M1 : M0, (nm - 1) >> 1);
                if (lane == 0) scarry[fl] = last;
                const float a = fminf(alpha[f], 1.0f);
                const float inv_r = 1.0f / fmaxf(root[f], 1.0f);
                const float dl = delta[f];
                // q = p / (floor+M)^a with the hardware log2/exp2 (1 ulp each; floor+M is a normal number).  Then
                // (q+d)^(1/r) - d^(1/r) = d^(1/r) * expm1(log1p(q/d)/r) for d > 0: no cancelling subtraction, so quiet
                // frames keep full relative accuracy and the general powf (the bulk of this kernel's arithmetic) is only
                // needed for d <= 0, where the reference's own formula is followed literally (NaN/inf cases included).
                const float q0 = v0 * leaf_pow_pos(floor_ + M0, -a);
                const float q1 = v1 * leaf_pow_pos(floor_ + M1, -a);
                if (dl > 0.0f) {
                    const float inv_d = 1.0f / dl;
                    r0 = s_dr[fl] * leaf_expm1_pos(inv_r * leaf_log1p_pos(q0 * inv_d));
                    r1 = s_dr[fl] * leaf_expm1_pos(inv_r * leaf_log1p_pos(q1 * inv_d));
                } else {
                    r0 = powf(q0 + dl, inv_r) - s_dr[fl];
                    r1 = powf(q1 + dl, inv_r) - s_dr[fl];
                }
            } else if (mode & 2) {
                r0 = log1pf(v0);
                r1 = log1pf(v1);
            }
            const size_t o = ((size_t)b * F + f) * TP + m0 + j0;
            if (mode & 4) {                                  // bf16 output, round to nearest even
                const unsigned u0 = __float_as_uint(r0), u1 = __float_as_uint(r1);
                if (j0 < nm) outh[o] = (unsigned short)((u0 + 0x7fffu + ((u0 >> 16) & 1u)) >> 16);
                if (j1 < nm) outh[o + 1] = (unsigned short)((u1 + 0x7fffu + ((u1 >> 16) & 1u)) >> 16);
            } else {
                if (j0 < nm) out[o] = r0;
                if (j1 < nm) out[o + 1] = r1;
            }
        }
        __syncthreads();
    }
}
#endif

// floor + optional log1p on an already pooled (B,F,T') tensor (staged path without PCEN)
#ifndef LEAF_INST_TU               // non-template kernel: compiled once, in leaf_kernels.hip
__global__ void floor_kernel(const float* __restrict__ p, size_t n, int mode, float* __restrict__ out) {
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= n) return;
    const float v = pooled_floor(p[idx]);
    out[idx] = (mode & 2) ? log1pf(v) : v;
}
#endif

}  // namespace
