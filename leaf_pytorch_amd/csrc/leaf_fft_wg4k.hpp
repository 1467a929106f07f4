// leaf_fft_wg4k.hpp -- overlap-save forward with 4096-sample blocks for the long 32 kHz window (K = 801, hop = 320)
// Part of the single translation unit leaf_kernels.hip (gfx950 only); see that file's header comment.
//
// Why: with 2048-sample blocks a K = 801 window leaves L = 960 valid outputs per transform (47 %); a 4096-sample block
// leaves 3200 (78 %).  A wave cannot hold a 4096-point transform (128 data registers), but it does not have to: one
// radix-2 decimation-in-frequency step splits the inverse transform into two INDEPENDENT 2048-point transforms -- the even
// output samples from zs[e] = Z[e] + Z[e + 2048], the odd ones from zd[e] = (Z[e] - Z[e + 2048]) w^e -- which the wave runs
// one after the other with the wave-level transform it already has (fft2048w).  The w^e twiddle is folded into a second
// pair of per-filter spectrum tables, so the step costs a few multiplies per bin, no extra pass:
//     zs[e] = conj(A'[e]) R[e] + A'[2048 - e] R[e + 2048]                  (A' real-input: conj(A'[e + 2048]) = A'[2048 - e])
//     zd[e] = conj(A'[e]) D_lo[e] - A'[2048 - e] D_hi[e],   D_lo = R[e] w^e,  D_hi = R[e + 2048] w^e,  w = e^{-2 pi i / 4096}
// Each half looks exactly like a 16 kHz block: 1600 valid samples at stride 2, pooled with the even / odd taps of the
// Gaussian window (401 / 400 taps, hop 160) -- the static pooling code of the 401/160 geometry, fed from de-interleaved
// pooling rows -- and both halves add into the same 13 frame accumulators.
// Round 4 (static 32 kHz kernel): (i) the odd half is computed as (conj(A'[e]) R[e] - A'[2048 - e] R[e + 2048]) w^e from the
// SAME real tables as the even half and ONE filter-independent twiddle table w^e (16 KB) -- the same eight instructions per
// bin as with the folded tables, but a task now touches 16 KB of per-filter tables instead of 48 KB: 80 filters are 1.3 MB
// instead of 3.9 MB next to a 4 MB L2 per XCD (396 MB through the fabric per launch against 102 MB algorithmic, round 3;
// the D tables are still built: the run-time-geometry and backward kernels read them); (ii) the pooling weights of a half
// live in 15 registers (the (row, frame) pairs of the 401/160 half-rate geometry share 15 distinct weight vectors,
// wg_pool_nj) instead of two wave-private LDS rows and a DMA per task: -160 ds_read_b32 per task, -8.4 KB of LDS per wave,
// which (iii) pays for the FULL transposition scratch: half the LDS store instructions of the column-half form.
// Everything else (workgroup per block, spectrum ring in LDS, task queue) is leaf_fft_wg.hpp; the forward task builds
// A' = FFT4096(block) from two 2048-point transforms of the even / odd input samples (decimation in time, combined
// through the ring slot itself).
#pragma once
#include "leaf_fft_wg.hpp"

namespace {

constexpr int kFft4N = 4096;
#ifndef LEAF_4K_FWD_NW
#define LEAF_4K_FWD_NW 12              // waves of the static 4096-sample forward kernel (A/B: 11)
#endif
#ifndef LEAF_SWEEP_BACK
#define LEAF_SWEEP_BACK 1              // 0: every block walks the filters 0 .. F - 1 (A/B)
#endif
constexpr int kWg4RingFloat2 = 2056;           // bins 0..2048 of a 4096-point spectrum, padded
constexpr int kWg4RowFloats = 528;             // one half pooling row: 64 zeros + 401 taps + 63 zeros (as the 401/160 geometry)
// static forward kernel (round 4): full transposition scratch per wave, no pooling rows (the weights live in registers)
constexpr size_t fft_wg4k_lds_bytes(int NW) {
    return ((size_t)kTwFloats + 2 * (32 + 64) + 2 * 2 * kWg4RingFloat2 + kWgQueueInts + (size_t)NW * kWgScrFloats) * 4;
}
// S801: the pooling backward reads the half's parity row as 13 register vectors per lane (the forward's form) instead of
// wave-private LDS rows filled by DMA: whole backward at cfg2 5.53 -> 5.41 ms, with dL/dx 7.01 -> 6.78 ms, same box
// (profiles/r04/ab_bwd_regw.txt).  0: the LDS rows (A/B).
#ifndef LEAF_4K_BWD_REGW
#define LEAF_4K_BWD_REGW 1
#endif
// ... which frees the rows' LDS: the parameter-gradient kernel then has room for the full transposition scratch of the forward
// (fewer LDS store instructions per transform): 5.58 -> 5.49 ms, same box.  0: the half-size scratch (A/B).
#ifndef LEAF_4K_BWD_FULLSCR
#define LEAF_4K_BWD_FULLSCR 1
#endif
#ifndef LEAF_4K_BWD_FUSE2
#define LEAF_4K_BWD_FUSE2 1            // S801 gradient rows in pairs (r, r + 16) with the following transform's first stage fused (-0.7 % of the cfg2 backward); 0: A/B
#endif
#ifndef LEAF_4K_BWD_PW2
#define LEAF_4K_BWD_PW2 1              // a second register set, the weights times (tap - centre)^2 (d pool_w); 0: squared per use (A/B)
#endif
constexpr int fft_wg4k_bwd_rows(bool dx) { return LEAF_4K_BWD_REGW ? 0 : dx ? 1 : 2; }   // wave-private pooling rows in LDS
// static BACKWARD kernel (leaf_fft_wgg4k_bwd_kernel<12, 7, true>) with the half-size transposition scratch (what it runs with
// LEAF_4K_BWD_FULLSCR = 0; with it, the forward's fft_wg4k_lds_bytes) and, LEAF_4K_BWD_REGW = 0, the filter's two parity rows
constexpr size_t fft_wg4k_bwd_lds_bytes(int NW) {
    return ((size_t)kTwFloats + 2 * (32 + 64) + 2 * 2 * kWg4RingFloat2 + kWgQueueInts +
            (size_t)NW * (kWgScrHalfFloats + fft_wg4k_bwd_rows(false) * kWg4RowFloats)) * 4;
}
// ... with dL/dx (leaf_fft_wgg4k_bwd_kernel<8, 7, true, true>): half-size scratch + three folded gradient spectra and their
// tickets.  Eight waves: two per SIMD with 256 VGPRs each -- the filter's R_lo / R_hi stay in
// registers for the task (nine waves at 168 VGPRs measured 11 % slower: profiles/r04/ab_4k_dx.txt)
#ifndef LEAF_4K_BWD_NW
#define LEAF_4K_BWD_NW 12            // waves of the static 4096-sample backward without dL/dx (A/B: 8 = two per SIMD, 256 VGPRs)
#endif
#ifndef LEAF_4K_DX_NW
#define LEAF_4K_DX_NW 8
#endif
constexpr int kWg4BwdDxWaves = LEAF_4K_DX_NW;
constexpr size_t fft_wg4k_bwd_dx_lds_bytes(int NW) {
    return ((size_t)kTwFloats + 2 * (32 + 64) + 2 * 2 * kWg4RingFloat2 + kWgQueueInts +
            (size_t)NW * (kWgScrHalfFloats + fft_wg4k_bwd_rows(true) * kWg4RowFloats)) * 4 + (size_t)3 * kWg4RingFloat2 * 8 + 64;
}
// the filter-independent twiddle table of the odd half, w^e = e^{-2 pi i e / 4096}, e < 2048 (float2), behind the pooling rows
constexpr size_t kFft4WtFloats = 2 * 2048;
// per-filter tables of the 4096-point plan (floats): (R_lo, R_hi)[2048] f2 | (D_lo, D_hi)[2048] f4   (derivative slabs: the second part
// holds (d/dmu lo, d/dmu hi, d/dsigma lo, d/dsigma hi)[2048] in the mu slab; fft4k_prep_kernel)
constexpr size_t kFft4TabFloats = 2048 * 6;

// ---- tables: one workgroup per filter.  z = conj(taps) in zero-phase layout over 4096 points; its spectrum through two
// 2048-point transforms (even / odd samples) and the decimation-in-time butterfly; real by the Hermitian symmetry of the
// taps about the centre (impulse_responses.py:5-16), 1 / 4096 folded in.
#ifndef LEAF_INST_TU               // non-template kernel: compiled once, in leaf_kernels.hip
__global__ __launch_bounds__(kPrepWaves * 64) void fft4k_prep_kernel(const float* __restrict__ kernel, const float* __restrict__ pool_w,
                                                                     int F, int K, GaborBounds bd, float* __restrict__ tab,
                                                                     float* __restrict__ Grow, int RG, float2* __restrict__ Wt = nullptr,
                                                                     const BandTabArgs a = BandTabArgs{}) {
    __shared__ float2 s_twl[32 * 64];
    __shared__ float2 s_twh[64];
    __shared__ float s_scr[32 * 65];
    __shared__ float2 s_taps[2048 + 64];                                  // conj(w_f), K <= 2049; afterwards the spectrum R[4096] (band decision)
    __shared__ float red[kPrepWaves][7];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int f = blockIdx.x;
    // blockIdx.y (backward tables): 0 the taps w, 1 d w / d mu = i t w, 2 d w / d sigma = (t^2 / s^3 - 1 / s) w -- as
    // fft_prep_kernel; slab `which` of the table holds their spectra (the D tables matter for the taps only)
    const int which = blockIdx.y;
    tab += (size_t)which * F * kFft4TabFloats;
    const float mu = kernel[2 * f], sg = kernel[2 * f + 1];
    const float sgc = fminf(fmaxf(sg, bd.sigma_lo), bd.sigma_hi);
    for (int j = tid; j < K; j += kPrepWaves * 64) {
        float a_, b_;
        const float t = (float)(j - K / 2);
        gabor_tap(mu, sg, bd, t, a_, b_);
        if (which == 1) {
            const float a0 = a_;
            a_ = -t * b_;
            b_ = t * a0;
        } else if (which == 2) {
            const float c = t * t / (sgc * sgc * sgc) - 1.0f / sgc;
            a_ *= c;
            b_ *= c;
        }
        s_taps[j] = make_float2(a_, -b_);
    }
    if (which == 0) {   // de-interleaved pooling rows: Ge[64 + i] = g[2 i], Go[64 + i] = g[2 i + 1]  (impulse_responses.py:74-80)
        const float half = 0.5f * (float)(K - 1);
        const float sp = pool_sigma(pool_w[f], K);
        for (int jj = tid; jj < 2 * RG; jj += kPrepWaves * 64) {             // RG floats per parity row (kWg4RowFloats for K = 801)
            const int h = jj / RG, i = jj - h * RG - kGPad, j = 2 * i + h;
            float v = 0.0f;
            if (i >= 0 && j < K) {
                const float q = ((float)j - half) / (sp * half);
                v = expf(-0.5f * (q * q));
            }
            Grow[(size_t)f * 2 * RG + jj] = v;
        }
    }
    if (which == 0 && f == 0 && Wt) {                                     // w^e, e < 2048: shared by every filter (round 4)
        for (int e = tid; e < 2048; e += kPrepWaves * 64) {
            float sn, cs;
            sincospif(2.0f * (float)e / (float)kFft4N, &sn, &cs);          // the expression the D tables are built from, below
            Wt[e] = make_float2(cs, -sn);
        }
    }
    fft_build_twiddles(s_twl, s_twh, tid, kPrepWaves * 64);
    __syncthreads();
    // band-limited filter tasks of the static forward kernel (leaf_band.hpp): the class decision is taken here, from the spectrum
    const bool decide = which == 0 && a.rec != nullptr;                   // (uniform over the workgroup)
    if (wave != 0 && !decide) return;
    float* Rs = reinterpret_cast<float*>(s_taps);                         // [4096]: entry i <-> R[i] (the filter lives at entries 4096 - bin)
    if (wave == 0) {
        auto tap_at = [&](int i) {                                        // zero-phase layout: index i <-> t = i (i < 2048) or i - 4096
            const int j = (i < kFft4N / 2 ? i : i - kFft4N) + K / 2;
            return (j >= 0 && j < K) ? s_taps[j] : make_float2(0.0f, 0.0f);
        };
        // the even samples' transform first, then the odd ones' (round 5: both sets loaded up front were 128 live registers next
        // to a transform's temporaries -- 244 B of scratch per lane)
        float ere[32], eim[32];
#pragma unroll
        for (int r = 0; r < 32; ++r) {
            const float2 te = tap_at(2 * (64 * r + lane));
            ere[r] = te.x; eim[r] = te.y;
        }
        fft2048(ere, eim, s_scr, s_twl, s_twh, lane);
        asm volatile("" : "+v"(ere[0]), "+v"(ere[31]) : : "memory");        // (the odd samples are read after it)
        float ore[32], oim[32];
#pragma unroll
        for (int r = 0; r < 32; ++r) {
            const float2 to = tap_at(2 * (64 * r + lane) + 1);
            ore[r] = to.x; oim[r] = to.y;
        }
        fft2048(ore, oim, s_scr, s_twl, s_twh, lane);
        // slab layout (kFft4TabFloats floats per filter): R2[2048] float2 = (R_lo[e], R_hi[e]) | D4[2048] float4 = (D_lo[e], D_hi[e]):
        // the values a bin needs together sit together, one 8- / 16-byte load per bin instead of two (round 4: a load instruction
        // costs these kernels more than four VALU instructions)
        float2* R2 = reinterpret_cast<float2*>(tab + (size_t)f * kFft4TabFloats);
        float4* D4 = reinterpret_cast<float4*>(tab + (size_t)f * kFft4TabFloats + 4096);
#pragma unroll
        for (int i = 0; i < 32; ++i) {
            const int e = 64 * brev5(i) + lane;
            float s, c;
            sincospif(2.0f * (float)e / (float)kFft4N, &s, &c);              // w^e = (c, -s)
            const float tr = ore[i] * c + oim[i] * s;                         // Re(w^e Xo[e])
            const float rlo = (ere[i] + tr) * (1.0f / kFft4N), rhi = (ere[i] - tr) * (1.0f / kFft4N);
            if (which == 0) {
                R2[e] = make_float2(rlo, rhi);
                D4[e] = make_float4(rlo * c, -rlo * s, rhi * c, -rhi * s);
                if (decide) { Rs[e] = rlo; Rs[e + 2048] = rhi; }          // (this wave was the taps' only reader)
            } else {
                // the derivative tables' D slots carry (d/dmu lo, d/dmu hi, d/dsigma lo, d/dsigma hi) of bin e as ONE 16-byte entry
                // (in the mu slab: which = 1 writes the first half of each entry, which = 2 the second): the backward's dot products
                // take one load per bin and half instead of four (-1.4 .. -2.1 % of the backward: profiles/r04/ab_table_addressing.txt)
                float2* ms = reinterpret_cast<float2*>(tab - (size_t)(which - 1) * F * kFft4TabFloats + (size_t)f * kFft4TabFloats + 4096);
                ms[2 * e + (which - 1)] = make_float2(rlo, rhi);
            }
        }
    }
    if (!decide) return;
    __syncthreads();
    // ---- class decision (leaf_band.hpp; one class here: a 512-bin window of the 4096-point spectrum, decimation 8): the window
    // [kb, kb + 512) around the centre bin inside bins 1..2048 are entries rlo .. rlo + 511 of R (descending bins)
    constexpr int M = 512;
    const float muc = fminf(fmaxf(mu, 0.0f), 3.14159274101257324f);
    const int k0 = (int)rintf(muc * (float)(kFft4N / 6.283185307179586));
    const int kb = min(max(k0 - M / 2, 1), kFft4N / 2 + 1 - M);
    const int rlo = kFft4N - kb - M + 1;
    float sums[4] = {0.0f, 0.0f, 0.0f, 0.0f};                            // total | outside the window | |autocorrelation| at M/2, 3M/4
    float mxo = 0.0f;                                                    // the largest dropped R^2 (round 6: the bias bound, leaf_band.hpp)
    float dcs = 0.0f, mxdc = 0.0f;                                       // ... and what is dropped at DC: bins 0, -1 .. -63 = entries 0 .. 63 (kBandAdjacentDC)
#pragma unroll
    for (int i0 = 0; i0 < kFft4N; i0 += kPrepWaves * 64) {
        const int i = i0 + tid;
        const float v = Rs[i];
        const int j = i - rlo;
        const bool in = j >= 0 && j < M;
        sums[0] += v * v;
        sums[1] += in ? 0.0f : v * v;
        mxo = fmaxf(mxo, in ? 0.0f : v * v);
        dcs += i < 64 ? v * v : 0.0f;
        mxdc = fmaxf(mxdc, i < 64 ? v * v : 0.0f);
        sums[2] += in && j + M / 2 < M ? fabsf(v * Rs[min(i + M / 2, kFft4N - 1)]) : 0.0f;
        sums[3] += in && j + 3 * M / 4 < M ? fabsf(v * Rs[min(i + 3 * M / 4, kFft4N - 1)]) : 0.0f;
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const float w = wave_sum(sums[k]);
        if (lane == 0) red[wave][k] = w;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mxo = fmaxf(mxo, __shfl_xor(mxo, o));
    if (lane == 0) red[wave][4] = mxo;
    {
        const float w = wave_sum(dcs);
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) mxdc = fmaxf(mxdc, __shfl_xor(mxdc, o));
        if (lane == 0) { red[wave][5] = w; red[wave][6] = mxdc; }
    }
    __syncthreads();
    if (tid == 0) {
        float v[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            float s = 0.0f;
#pragma unroll
            for (int w = 0; w < kPrepWaves; ++w) s += red[w][k];
            v[k] = s;
        }
        float odc = 0.0f, mdc = 0.0f;
#pragma unroll
        for (int w = 0; w < kPrepWaves; ++w) { odc += red[w][5]; mdc = fmaxf(mdc, red[w][6]); }
        const float o2 = a.eta < kBandEta ? v[1] + (kBandAdjacentDC / kBandAdjacent - 1.0f) * odc : v[1];   // (leaf_band.hpp: what is dropped at DC)
        bool ok = o2 <= a.eps2 * v[0] && v[2] <= a.eta * v[0] && v[3] <= a.eta * v[0];
        // the smallest bias that admits the class beyond the strict rule (leaf_band.hpp: band_need, band_pool_gamma)
        const float sk = (float)kFft4N / (6.2831853f * sgc), spw = pool_sigma(pool_w[f], K);
        const float dmin = (float)min(k0 - (kb - 1), kb + M - k0) - 2.0f * sk;
        float mx = 0.0f;
#pragma unroll
        for (int w = 0; w < kPrepWaves; ++w) mx = fmaxf(mx, red[w][4]);
        int nd = band_need(v[1], mx, v[2], v[3], v[0], a.eta, fabsf(Rs[(kFft4N - k0) & (kFft4N - 1)]),
                           band_pool_gamma(spw, K, dmin, kFft4N), spw, K, kFft4N, 0.0f, M, odc, mdc);
        if (a.bwd_slabs && !band_deriv_fits(k0, kb, M, sk)) { ok = false; nd = kBandNever; }          // (backward: leaf_band.hpp, kBandDerivCore)
        if (a.force) { ok = a.force == 2; nd = kBandNever; }
        if (a.classes) a.classes[f] = (ok || band_bias_admits(a.cls_bias, f, true, nd)) ? M : kFft4N;
        a.rec[4 * f] = ok ? 2 : 0;                                        // (bit 1: the four-filters-per-task class of band_build_plan)
        a.rec[4 * f + 1] = kb;
        a.rec[4 * f + 2] = kb;
        a.rec[4 * f + 3] = kBandNever | (nd << 16);                       // bmin (fp16 codes): never the eight-per-task class | the 512-bin class
    }
}

// The tables of the band tasks on 4096-sample blocks (K = 801 / hop = 320; as the blockIdx.y > 0 workgroups of
// fft_prep_band_kernel, one class).  Grid (F, 1 + n_edge): workgroup (f, 0) the decimated pooling window G~(tau) =
// 8 sum_u g[tau - u] phi_8[|u|], tau = c0min + 8 j; workgroup (f, 1 + s) the dense table of edge entry s over the block's 512
// decimated samples, in the register order of the task (the lane reads entry 16 k + l2 of its register k).
__global__ __launch_bounds__(kPrepWaves * 64) void fft4k_band_tab_kernel(const float* __restrict__ pool_w, int F, int K, const BandTabArgs a) {
    __shared__ float gs[64 * kPoolRowsMax];
    __shared__ float gs2[64 * kPoolRowsMax];
    __shared__ float phis[kBandLh * 8 + 1];
    __shared__ int es[kBandMaxEdge][4];
    constexpr int D = 8, LPHI = kBandLh * D;
    const int tid = threadIdx.x, f = blockIdx.x, l16 = tid & 15, grp = tid >> 4;
    if (tid <= LPHI) phis[tid] = kBandPhi8[tid];
#pragma unroll
    for (int s = 0; s < kBandMaxEdge; ++s)
        if (tid == 256 + s) { es[s][0] = a.e[s].c; es[s][1] = a.e[s].m; es[s][2] = a.e[s].lo; es[s][3] = a.e[s].hi; }
    {
        const float half = 0.5f * (float)(K - 1), sp = pool_sigma(pool_w[f], K);
        for (int j = tid; j < K; j += kPrepWaves * 64) {                   // the pooling window, as fft4k_prep_kernel evaluates it
            const float q = ((float)j - half) / (sp * half);
            gs[j] = expf(-0.5f * (q * q));
            gs2[j] = gs[j] * (((float)j - half) * ((float)j - half));     // (backward tables: d pool_w, leaf_band_bwd.hpp)
        }
    }
    __syncthreads();
    auto entry = [&](int p0, int lo, int hi, int goff, const float* win) {   // sixteen lanes per table entry
        float acc = 0.0f;
#pragma unroll 4
        for (int pp = lo + l16; pp <= hi; pp += 16) {
            const int u = p0 - pp;
            acc = fmaf(win[pp + goff], phis[u < 0 ? -u : u], acc);
        }
        return band_row_sum(acc);
    };
    if (blockIdx.y == 0) {
        const int len = band_gz_len_d(K, a.hop, 32, D), c0 = band_c0min_d(K, a.hop, 32, D);
        const int gzs = band4k_gz_floats(K, a.hop);                        // (run-time gcd loops: once, not per entry)
        float* gzf = a.gz + (size_t)f * gzs;
        float* gz2f = a.gz2 ? a.gz2 + (size_t)f * gzs : nullptr;
        for (int j = grp; j < len; j += kPrepWaves * 4) {
            const int tau = c0 + D * j;
            const float v = entry(tau, max(0, tau - LPHI), min(K - 1, tau + LPHI), 0, gs);
            if (l16 == 0) gzf[j] = (float)D * v;
            if (a.gz2) {
                const float v2 = entry(tau, max(0, tau - LPHI), min(K - 1, tau + LPHI), 0, gs2);
                if (l16 == 0) gz2f[j] = (float)D * v2;
            }
        }
        if (f == 0 && tid < 4 * kBandMaxEdge) a.elist[tid] = es[tid >> 2][tid & 3];
        return;
    }
    const int s = blockIdx.y - 1;
    const int c = es[s][0], goff = c * a.L - (es[s][1] * a.hop - a.padL);
    const int pa = es[s][2] - c * a.L, pb = es[s][3] - c * a.L;
    float* tabe = a.edge + ((size_t)f * kBandMaxEdge + s) * 512;
    for (int m = grp; m < 512; m += kPrepWaves * 4) {
        int p0 = m * D;
        if (p0 - kFft4N + LPHI >= pa) p0 -= kFft4N;
        else if (p0 + kFft4N - LPHI < pb) p0 += kFft4N;
        const float v = entry(p0, max(pa, p0 - LPHI), min(pb - 1, p0 + LPHI), goff, gs);
        if (l16 == 0) tabe[brev5(m >> 4) * 16 + (m & 15)] = (float)D * v;
        if (a.edge2) {
            const float v2 = entry(p0, max(pa, p0 - LPHI), min(pb - 1, p0 + LPHI), goff, gs2);
            if (l16 == 0) a.edge2[(tabe - a.edge) + brev5(m >> 4) * 16 + (m & 15)] = (float)D * v2;
        }
    }
}
#endif

// Rows k = 0..31 of the two ring streams a 4096-point multiply needs, eight rows at a time:
//   a[j] = A'[64 k + lane],  m[j] = A'[2048 - 64 k - lane],  k = 8 C + j.
template <int C>
__device__ __forceinline__ void wg4k_ring_chunk(v2f (&a)[8], v2f (&m)[8], unsigned a_dir, unsigned a_mir) {
    lds_rd8<512 * (8 * C + 0)>(a[0], a_dir); lds_rd8<512 * (8 * C + 1)>(a[1], a_dir); lds_rd8<512 * (8 * C + 2)>(a[2], a_dir);
    lds_rd8<512 * (8 * C + 3)>(a[3], a_dir); lds_rd8<512 * (8 * C + 4)>(a[4], a_dir); lds_rd8<512 * (8 * C + 5)>(a[5], a_dir);
    lds_rd8<512 * (8 * C + 6)>(a[6], a_dir); lds_rd8<512 * (8 * C + 7)>(a[7], a_dir);
    // mirror: byte address base + 512 (31 - k), base = &A'[2048 - 64 * 31 - lane]
    lds_rd8<512 * (31 - (8 * C + 0))>(m[0], a_mir); lds_rd8<512 * (31 - (8 * C + 1))>(m[1], a_mir);
    lds_rd8<512 * (31 - (8 * C + 2))>(m[2], a_mir); lds_rd8<512 * (31 - (8 * C + 3))>(m[3], a_mir);
    lds_rd8<512 * (31 - (8 * C + 4))>(m[4], a_mir); lds_rd8<512 * (31 - (8 * C + 5))>(m[5], a_mir);
    lds_rd8<512 * (31 - (8 * C + 6))>(m[6], a_mir); lds_rd8<512 * (31 - (8 * C + 7))>(m[7], a_mir);
    asm volatile("s_waitcnt lgkmcnt(0)"
                 : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]),
                   "+v"(m[0]), "+v"(m[1]), "+v"(m[2]), "+v"(m[3]), "+v"(m[4]), "+v"(m[5]), "+v"(m[6]), "+v"(m[7]));
}

#ifndef LEAF_4K_FWD_WT
#define LEAF_4K_FWD_WT 0               // 1: the odd half's twiddles w^e from the shared global table (32 loads per task) instead of w^(64 k) w^lane from the LDS tables: 1.4 % slower at cfg2 (A/B)
#endif
// A table load as  uniform base (SGPR pair) + this lane's byte offset (one VGPR, zero-extended) + a compile-time byte offset:
// the form global_load takes without any address arithmetic in the VALU.  `voff` is what call sites make opaque to pin a group
// of loads in place (an opaque element INDEX costs ~3 VALU instructions of 64-bit arithmetic per load; opaque 64-bit pointers
// cost register pairs).
template <typename T = float>
__device__ __forceinline__ T tab_ld(const void* ubase, unsigned voff, int const_bytes) {
    return *reinterpret_cast<const T*>(static_cast<const char*>(ubase) + (size_t)voff + const_bytes);
}

template <int SK, int SHOP, int NW>
__global__ __launch_bounds__(NW * 64, (NW + 3) / 4) void leaf_fft_wg4k_kernel(const FftParams p) {
    static_assert(SK == 801 && SHOP == 320, "the 4096-sample plan is instantiated for the 32 kHz LEAF geometry");
    using gfp = const __attribute__((address_space(1))) float*;          // table pointers that stay `global` when made opaque
    using gf2p = const __attribute__((address_space(1))) v2f*;
    constexpr int SCRF = kWgScrFloats;                                    // full transposition scratch (round 4: no pooling rows in LDS)
    extern __shared__ __attribute__((aligned(16))) float wsm[];
    float2* twl = reinterpret_cast<float2*>(wsm);                        // [32][64]
    float2* twh = twl + 32 * 64;                                          // [32][2]
    float2* tw4a = twh + 64;                                              // w^(64 k), k < 32
    float2* tw4b = tw4a + 32;                                             // w^lane
    float2* ring = tw4b + 64;                                             // [2][kWg4RingFloat2]
    int* q = reinterpret_cast<int*>(ring + 2 * kWg4RingFloat2);
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane0 = tid & 63;
    float* scr = reinterpret_cast<float*>(q + kWgQueueInts) + (size_t)wave * SCRF;
    const unsigned scr_lds = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(__attribute__((address_space(3))) float*)scr);

    // band-limited filter tasks (leaf_band.hpp, round 5): the plan sits behind the waves' scratch
    const bool band_on = p.band.rec != nullptr;
    int* bl = reinterpret_cast<int*>(wsm + p.band.lds_off);
    if (band_on && wave == 0) band_build_plan(p.band.rec, p.band.elist, p.band.n_edge, p.F, bl, lane0, p.band.bias, p.band.smax);

    if (band_on) { if (wave > 0) fft_build_twiddles_wg(twl, twh, tid - 64, (NW - 1) * 64); }   // (wave 0 builds the plan meanwhile)
    else fft_build_twiddles_wg(twl, twh, tid, NW * 64);
    for (int i = tid; i < 96; i += NW * 64) {
        float s, c;
        sincospif(2.0f * (float)(i < 32 ? 64 * i : i - 32) / (float)kFft4N, &s, &c);
        tw4a[i] = make_float2(c, -s);                                     // (tw4b follows tw4a contiguously)
    }
    if (tid < kWgQueueInts) q[tid] = 0;
    __syncthreads();

    // full-rate geometry of the block and the half-rate geometry both halves share (= the 401 / 160 static geometry)
    constexpr int PADL = SK / 2 + SK % 2 - 1;                             // 400
    constexpr int LS = 3200;                                              // valid outputs per 4096-sample block (10 hops, 50 rows)
    constexpr int HK = (SK + 1) / 2, HHOP = SHOP / 2, HPADL = PADL / 2;   // 401, 160, 200
    constexpr int HLS = LS / 2;                                           // 1600 samples per half
    constexpr int DMIN = -((HK - 1 - HPADL) / HHOP);
    constexpr int DMAX = (HLS - 1 + HPADL) / HHOP;
    constexpr int NFR = DMAX - DMIN + 1;
    constexpr int NROW = HLS / 64;
    static_assert(NFR <= 16 && NROW == 25 && LS % SHOP == 0 && kFft4N - SK + 1 >= LS, "4096-sample plan geometry");
    // pooling weights of a half in registers: the half-rate geometry is the 401 / 160 one (wg_pool_nj: 15 vectors at stride 32)
    constexpr int PG = wg_pool_step(HHOP), PJ0 = wg_pool_jmin(HK, HHOP), NJ = wg_pool_nj(HK, HHOP);
    static_assert((HPADL - PJ0) % PG == 0, "window offsets are congruent to padL modulo gcd(64, hop)");
    const float2* Wt = reinterpret_cast<const float2*>(p.lone);           // w^e, e < 2048 (fft4k_prep_kernel; p.lone carries it here)

    // blocks dealt contiguously; clips all of whose blocks this workgroup ran are finalized in its tail (as leaf_fft_wg_kernel)
    const OwnedClips deal{p.B * p.nblk, (int)gridDim.x, p.nblk};
    const int first_gb = deal.start((int)blockIdx.x);
    const int nset = deal.count((int)blockIdx.x);
    // filter tasks per block: one per filter, or (band tasks) one per wide filter + one per four narrow-band filters
    const int NT = band_on ? __builtin_amdgcn_readfirstlane(bl[0]) : p.F;
    const int* tdesc = bl + kBandPlanHead;
    const int* bmem = tdesc + p.F + 4;
    const WgTaskGrid grid = wg_task_grid(NT, nset);                        // NT + 1 slots per set
    const int ntasks = nset > 0 ? 1 + nset * (NT + 1) : 0;
    auto pull = [&]() {
        int v = 0;
        if (lane0 == 0) v = __hip_atomic_fetch_add(&q[0], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        return __builtin_amdgcn_readfirstlane(v);
    };
    auto decode = [&](int t, int& set, int& role) { wg_task_decode(grid, t, set, role); };

    int seen_set = -1, seen_b = 0, seen_c = 0;                            // block coordinates of the set this wave last worked on
    int t = pull(), set = 0, role = 0;
    if (t < ntasks) decode(t, set, role);
    while (t < ntasks) {
        int lane = lane0;
        asm volatile("" : "+v"(lane));
        const int slot = set & 1, gen = set >> 1;
        float2* A = ring + slot * kWg4RingFloat2;
        if (role == 0 || role > NT) {
            if (role == 0 && set < nset) {
                // ---- A' = FFT4096(rotated block), bins 0..2048, by decimation in time: Xe = FFT2048(even samples) parked in
                // the ring slot, Xo = FFT2048(odd samples), A'[e] = Xe[e] + w^e Xo[e], A'[2048] = Xe[0] - Xo[0]
                const int gb = first_gb + set;
                const int b = gb / p.nblk, c = gb - b * p.nblk;
                const int n_c = c * LS;
                const float* xb = static_cast<const float*>(p.x) + (size_t)b * p.T;
                const unsigned short* xh = static_cast<const unsigned short*>(p.x) + (size_t)b * p.T;
                auto sample = [&](int i) -> float {                       // rotated block a'[i] = xz[n_c - padL + ((i + padL) mod 4096)]
                    const int n = n_c - PADL + ((i + PADL) & (kFft4N - 1));
                    if (p.io_bf16) {
                        const unsigned v = xh[min(max(n, 0), p.T - 1)];
                        return (n >= 0 && n < p.T) ? __uint_as_float(v << 16) : 0.0f;
                    }
                    return (n >= 0 && n < p.T) ? xb[n] : 0.0f;
                };
                float xre[32], xim[32];
#pragma unroll
                for (int r = 0; r < 32; ++r) { xre[r] = sample(2 * (64 * r + lane)); xim[r] = 0.0f; }
                fft2048w<false>(xre, xim, scr, scr_lds, twl, twh, lane);
                wg_wait_ge(&q[3 + slot], gen * NT);                       // the slot's previous readers are done
#pragma unroll
                for (int i = 0; i < 32; ++i) A[64 * brev5(i) + lane] = make_float2(xre[i], xim[i]);
#pragma unroll
                for (int r = 0; r < 32; ++r) { xre[r] = sample(2 * (64 * r + lane) + 1); xim[r] = 0.0f; }
                fft2048w<false>(xre, xim, scr, scr_lds, twl, twh, lane);
                const float2 wl = tw4b[lane];
#pragma unroll
                for (int i = 0; i < 32; ++i) {
                    const int k = brev5(i);
                    const float2 wk = tw4a[k];
                    const float wr = wk.x * wl.x - wk.y * wl.y, wi = wk.x * wl.y + wk.y * wl.x;      // w^(64 k + lane)
                    const float tr = xre[i] * wr - xim[i] * wi, ti = xre[i] * wi + xim[i] * wr;
                    const float2 xe = A[64 * k + lane];
                    A[64 * k + lane] = make_float2(xe.x + tr, xe.y + ti);
                    if (k == 0 && lane == 0) A[2048] = make_float2(xe.x - tr, xe.y - ti);
                }
                if (lane == 0) { q[5 + 2 * slot] = b; q[6 + 2 * slot] = c; }
                wg_release();
                if (lane == 0) __hip_atomic_fetch_add(&q[1 + slot], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
            t = pull();
            if (t < ntasks) decode(t, set, role);
            continue;
        }
        // ---- filter task of the block in ring slot `slot`.  Odd sets walk the tasks backwards: the per-filter tables (16 KB each
        // for this kernel; 1.3 MB at 80 filters, next to a 4 MB L2 per XCD) are swept once per block by every workgroup, and a
        // sweep that turns around re-reads the tables it used last while they are still resident instead of evicting them in order
        const int ti = (LEAF_SWEEP_BACK && (set & 1)) ? NT - role : role - 1;
        const int tdsc = band_on ? __builtin_amdgcn_readfirstlane(tdesc[ti]) : ti << 2;   // class (0: one filter, two 2048-point halves; 2: band task) | index << 2
        const int f = tdsc >> 2;
        if (set != seen_set) {                                            // this wave's first filter of the block: once the
            wg_wait_ge(&q[1 + slot], gen + 1);                            // spectrum is in the ring it stays until every filter is done
            seen_b = __builtin_amdgcn_readfirstlane(wg_ld(&q[5 + 2 * slot]));
            seen_c = __builtin_amdgcn_readfirstlane(wg_ld(&q[6 + 2 * slot]));
            seen_set = set;
        }
        const int b = seen_b, c = seen_c;
        const int n_c = c * LS;
        const int Lv = min(LS, p.T - n_c);
        int mlo = n_c + PADL - SK + 1;
        mlo = mlo <= 0 ? 0 : (mlo + SHOP - 1) / SHOP;
        const int mhi = min(p.TP - 1, (n_c + Lv - 1 + PADL) / SHOP);
        if (tdsc & 3) {
            // ---- band task: four narrow-band filters on 512-point transforms of their windows of the 4096-point spectrum
            // (decimation 8; leaf_band.hpp).  The filter's values for bins kb + j are R[4096 - kb - j] = R_hi[2048 - kb - j].
            const int* mem = bmem + (tdsc >> 2);
            float rq[32];
            {
                const int me1 = mem[lane / band_lpf(32)];
                const int fid = me1 & 0xffff, kb = (me1 >> 16) & 0xfff, c1 = lane & (band_lpf(32) - 1);
                const float* src = reinterpret_cast<const float*>(p.H) + (size_t)fid * kFft4TabFloats + 2 * (2048 - kb - c1) + 1;
                asm volatile("" ::: "memory");
#pragma unroll
                for (int k = 0; k < 32; ++k) rq[k] = src[-2 * (32 * (k & 15) + 16 * (k >> 4))];
                asm volatile("" ::: "memory");
            }
            int tn_b = 0, nset_b = 0, nrole_b = 0;
            auto mid = [&]() {
                tn_b = pull();
                if (tn_b < ntasks) decode(tn_b, nset_b, nrole_b);
                return 0;                                                 // no table loads for the next task
            };
            auto stamp = [&](int) {};
            auto bout = [&](int fid, int m, float v) {                    // the block's share of frame m: where the full task puts it
                const int first_block = max(0, m * SHOP - PADL) / LS;
                p.part[(((size_t)b * p.F + fid) * p.nslot + (c - first_block)) * p.TP + m] = v;
            };
            band_task<32, SK, SHOP, true>(p, rq, A, mem, bl + 4, twl, scr, scr_lds, &q[3 + slot], c, mlo, mhi, lane, mid, bout, stamp);
            t = tn_b;
            set = nset_b;
            role = nrole_b;
            continue;
        }
        const float* Rtab = reinterpret_cast<const float*>(p.H) + (size_t)f * kFft4TabFloats;   // R_lo[2048] | R_hi[2048] of this filter (wave-uniform)
        const unsigned lane4 = 4u * (unsigned)lane;
        const float* gsrc = p.Gz + (size_t)f * 2 * kWg4RowFloats + (kGPad + PJ0) + lane;   // de-interleaved pooling rows (even | odd taps)
        const unsigned a_dir = lds_addr(A + lane), a_mir = lds_addr(A + (2048 - 64 * 31) - lane);
        float zre[32], zim[32];
        // pooling of one half: sample j of the half (register i <-> j = 64 brev5(i) + lane) is output n_c + 2 j + h
        // (returns the wave-reduced frame sums of this half: after the butterfly every lane holds the total of frame
        // fi(lane); one register carried across the other half instead of sixteen accumulators)
        auto pool_half = [&](auto hh, const float (&pw)[NJ]) -> float {
            constexpr int h = decltype(hh)::value;
            float acc[16];
#pragma unroll
            for (int fi = 0; fi < 16; ++fi) acc[fi] = 0.0f;
            float er[NROW];
#pragma unroll
            for (int i = 0; i < 32; ++i) {
                const int r = brev5(i);
                if (r < NROW) er[r] = zre[i] * zre[i] + zim[i] * zim[i];
            }
            if (Lv < LS) {
#pragma unroll
                for (int r = 0; r < NROW; ++r) er[r] = 2 * (64 * r + lane) + h < Lv ? er[r] : 0.0f;
            }
#pragma unroll
            for (int r = 0; r < NROW; ++r) {
#pragma unroll
                for (int fi = 0; fi < NFR; ++fi) {
                    const int is = (DMIN + fi) * HHOP - HPADL;            // first half-rate sample of frame fi's window
                    if (is <= 64 * r + 63 && is + HK > 64 * r) acc[fi] = fmaf(er[r], pw[(64 * r - is - PJ0) / PG], acc[fi]);
                }
            }
            return frame_butterfly16(acc, lane);
        };
        // ---- even output samples: zs = conj(A'[e]) R_lo[e] + A'[2048 - e] R_hi[e]
        {
            auto chunk = [&](auto cc) {
                constexpr int C = decltype(cc)::value;
                float rl[8], rh[8];
                // the table offset is made opaque HERE: the loads below cannot issue before this point (a plain "memory"
                // clobber does not hold them -- they are hoisted under the previous phase and spilled one by one)
                // (the chunk's table POINTERS, in the global address space so that the loads stay global_load: its eight loads per
                // table then differ by an immediate offset only -- an opaque index cost ~3 VALU instructions of 64-bit address
                // arithmetic per load)
                unsigned vo = lane4;
                if constexpr (C > 0) asm volatile("" : "+v"(vo), "+v"(zre[8 * C - 1]), "+v"(zim[8 * C - 1]) : : "memory");
                else asm volatile("" : "+v"(vo) : : "memory");
#pragma unroll
                for (int j = 0; j < 8; ++j) { const v2f r = tab_ld<v2f>(Rtab, 2u * vo, 512 * (8 * C + j)); rl[j] = r.x; rh[j] = r.y; }
                asm volatile("" ::: "memory");
                v2f a[8], m[8];
                wg4k_ring_chunk<C>(a, m, a_dir, a_mir);
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const int k = 8 * C + j;
                    zre[k] = fmaf(m[j].x, rh[j], a[j].x * rl[j]);
                    zim[k] = fmaf(m[j].y, rh[j], -(a[j].y * rl[j]));
                }
                asm volatile("" : "+v"(zre[8 * C]), "+v"(zre[8 * C + 1]), "+v"(zre[8 * C + 2]), "+v"(zre[8 * C + 3]),
                                  "+v"(zre[8 * C + 4]), "+v"(zre[8 * C + 5]), "+v"(zre[8 * C + 6]), "+v"(zre[8 * C + 7]),
                                  "+v"(zim[8 * C]), "+v"(zim[8 * C + 1]), "+v"(zim[8 * C + 2]), "+v"(zim[8 * C + 3]),
                                  "+v"(zim[8 * C + 4]), "+v"(zim[8 * C + 5]), "+v"(zim[8 * C + 6]), "+v"(zim[8 * C + 7]));
            };
            chunk(std::integral_constant<int, 0>{}); chunk(std::integral_constant<int, 1>{});
            chunk(std::integral_constant<int, 2>{}); chunk(std::integral_constant<int, 3>{});
        }
        // this half's pooling weights: requested now, they land under the transform
        float pw[NJ];
        auto load_weights = [&](int h) {
            asm volatile("" ::: "memory");
#pragma unroll
            for (int k = 0; k < NJ; ++k) pw[k] = gsrc[h * kWg4RowFloats + PG * k];
            asm volatile("" ::: "memory");
        };
        load_weights(0);
        fft2048w<false>(zre, zim, scr, scr_lds, twl, twh, lane);
        pin32(zre);
        pin32(zim);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                  // the weights have landed
        float v_even = pool_half(std::integral_constant<int, 0>{}, pw);
        // the first half's pooling is complete before the second half's table loads issue (else they are hoisted under it
        // and spilled one by one)
        asm volatile("" : "+v"(v_even) : : "memory");
        // ---- odd output samples: zd = (conj(A'[e]) R_lo[e] - A'[2048 - e] R_hi[e]) w^e
        {
            [[maybe_unused]] const float2 wl_odd = tw4b[lane];
            auto step = [&](auto cc) {                                    // four rows at a time (registers): k = 4 C4 .. 4 C4 + 3
                constexpr int C4 = decltype(cc)::value;
                unsigned vo = lane4;
                if constexpr (C4 > 0) asm volatile("" : "+v"(vo), "+v"(zre[4 * C4 - 1]), "+v"(zim[4 * C4 - 1]) : : "memory");
                else asm volatile("" : "+v"(vo) : : "memory");
                float rl[4], rh[4];
                v2f w[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const v2f r = tab_ld<v2f>(Rtab, 2u * vo, 512 * (4 * C4 + j));
                    rl[j] = r.x;
                    rh[j] = r.y;
#if LEAF_4K_FWD_WT
                    w[j] = tab_ld<v2f>(Wt, 2 * vo, 512 * (4 * C4 + j));
#else
                    {
                        const float2 wk = tw4a[4 * C4 + j];               // w^(64 k + lane) = w^(64 k) w^lane, both in LDS
                        w[j].x = wk.x * wl_odd.x - wk.y * wl_odd.y;
                        w[j].y = wk.x * wl_odd.y + wk.y * wl_odd.x;
                    }
#endif
                }
                asm volatile("" ::: "memory");
                v2f a[4], m[4];
                lds_rd8<512 * (4 * C4 + 0)>(a[0], a_dir); lds_rd8<512 * (4 * C4 + 1)>(a[1], a_dir);
                lds_rd8<512 * (4 * C4 + 2)>(a[2], a_dir); lds_rd8<512 * (4 * C4 + 3)>(a[3], a_dir);
                lds_rd8<512 * (31 - (4 * C4 + 0))>(m[0], a_mir); lds_rd8<512 * (31 - (4 * C4 + 1))>(m[1], a_mir);
                lds_rd8<512 * (31 - (4 * C4 + 2))>(m[2], a_mir); lds_rd8<512 * (31 - (4 * C4 + 3))>(m[3], a_mir);
                asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(m[0]), "+v"(m[1]),
                                                      "+v"(m[2]), "+v"(m[3]));
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int k = 4 * C4 + j;
                    // P = conj(a) rl - m rh = (ur, -ui);  zd = P w:  Re = ur w.x + ui w.y,  Im = ur w.y - ui w.x
                    const float ur = fmaf(-m[j].x, rh[j], a[j].x * rl[j]);
                    const float ui = fmaf(m[j].y, rh[j], a[j].y * rl[j]);
                    zre[k] = fmaf(ui, w[j].y, ur * w[j].x);
                    zim[k] = fmaf(ur, w[j].y, -(ui * w[j].x));
                }
                // every product of this step is complete before the next step's loads issue (VALU work may otherwise sink
                // below later volatile statements, keeping several steps' operands alive at once)
                asm volatile("" : "+v"(zre[4 * C4]), "+v"(zre[4 * C4 + 1]), "+v"(zre[4 * C4 + 2]), "+v"(zre[4 * C4 + 3]),
                                  "+v"(zim[4 * C4]), "+v"(zim[4 * C4 + 1]), "+v"(zim[4 * C4 + 2]), "+v"(zim[4 * C4 + 3]));
            };
            step(std::integral_constant<int, 0>{}); step(std::integral_constant<int, 1>{});
            step(std::integral_constant<int, 2>{}); step(std::integral_constant<int, 3>{});
            step(std::integral_constant<int, 4>{}); step(std::integral_constant<int, 5>{});
            step(std::integral_constant<int, 6>{}); step(std::integral_constant<int, 7>{});
        }
        wg_release();
        if (lane == 0) __hip_atomic_fetch_add(&q[3 + slot], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);   // ring reads done
        load_weights(1);                                                  // the odd taps' weights, under the second transform
        fft2048w<false>(zre, zim, scr, scr_lds, twl, twh, lane);
        pin32(zre);
        pin32(zim);
        const int tn = pull();                                            // next task reserved under the pooling
        int nset_i = 0, nrole = 0;
        if (tn < ntasks) decode(tn, nset_i, nrole);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                  // the weights have landed
        const float v_odd = pool_half(std::integral_constant<int, 1>{}, pw);
        {
            const float v = v_even + v_odd;
            const int fi = ((lane >> 5) & 1) * 8 + ((lane >> 4) & 1) * 4 + ((lane >> 3) & 1) * 2 + ((lane >> 2) & 1);
            const int m = n_c / SHOP + DMIN + fi;
            if ((lane & 3) == 0 && fi < NFR && m >= mlo && m <= mhi) {
                const int first_block = max(0, m * SHOP - PADL) / LS;
                p.part[(((size_t)b * p.F + f) * p.nslot + (c - first_block)) * p.TP + m] = v;
            }
        }
        t = tn;
        set = nset_i;
        role = nrole;
    }
    if (p.fin_fused)                                                      // the waves' rows are free: tile memory of the tail
        wg_tail_finalize<64>(p.fin, (first_gb + p.nblk - 1) / p.nblk, (first_gb + nset) / p.nblk,
                             reinterpret_cast<float*>(q + kWgQueueInts), tid, (int)blockDim.x);
}

}  // namespace
