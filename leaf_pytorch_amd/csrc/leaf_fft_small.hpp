// leaf_fft_small.hpp -- the whole forward of a SMALL batch in ONE launch (LEAF_ALGO_FFT_SMALL)
// One of the kernel families of libleaf_hip.so (gfx950 only); instantiated in inst_fft_small.hip, see leaf_inst.hpp.
//
// Why: inference in the reference is a handful of 1 s chunks per call (test.py:57-71,125-128: pad_input -> (n_sec,1,sr) ->
// mean over chunks).  At B = 4 the overlap-save path is 40 blocks of work on 256 CUs and three DEPENDENT launches -- tables
// (fft_prep_kernel), main kernel, row kernel: 31.8 us of which ~25 are launch latency and idle CUs (profiles/r03).  Here the
// three phases live in one kernel, and the unit of work is chosen so that no phase ever crosses a workgroup:
//
//     workgroup = (clip b, filter f), grid = (F, B), 11 waves
//
//   phase 0  all waves: twiddle tables, the filter's K taps (impulse_responses.py:5-16) into LDS, the clip's frame sums zeroed
//   phase 1  wave w < nblk: load + forward transform of block w of the clip -> its spectrum stays in the wave's registers
//            (round 5; rounds 3-4 parked the half-spectrum in an LDS ring: 8.2 KB per block, 97 LDS instructions per block);
//            the last wave meanwhile transforms the taps: R_f (the real spectrum of the zero-phase taps) -> LDS
//   phase 2  wave w < nblk: conj(A'_w) R_f in registers (the transform's output order is its input order bit-reversed: a
//            compile-time renaming) -> inverse transform -> |.|^2 -> Gaussian pooling with the filter's NJ weight vectors
//            (evaluated once per workgroup in phase 0, 15 x 64 expf spread over all threads, read from LDS) -> the frame
//            sums, ds_add_f32 into LDS (a window meets two blocks; a + b is the slot sum of the other kernels whichever way round)
//   (phases 1 and 2 share ONE copy of the wave-level transform in a two-trip loop: the code runs once per launch from a cold
//   instruction cache, and its size is most of the kernel's time -- see the loop)
//   phase 3  bias, floor, the EMA recurrence and PCEN of the row (b, f) by one wave: the fin_* point functions of leaf_fft.hpp;
//            the recurrence as a six-step lane scan (round 5: wave_affine_scan below) -- the smoothed value agrees with the other
//            kernels' sequential fin_ema_step chain to ~1e-7 relative, not to the bit
//
// A clip's forward transforms are repeated by each of the F workgroups that serve it.  That is 2x the arithmetic of the
// workgroup kernel -- on a chip that is 85 % idle at these sizes; what it buys is that the 40 filters of a clip run on 40 CUs
// at once, each one transform-time deep per phase: ~3 phases x ~2.5 us instead of 3 launches.  Clips longer than the ring
// (10 blocks) go through phases 1-2 in passes.  Results: the arithmetic of the workgroup kernels (fft2048w, mirrored upper
// half-spectrum, register pooling weights); tables differ from fft_prep_kernel's only in the rounding of the transform
// (~1e-7 relative).  A clip is bit-identical across every batch THIS kernel serves.
//
// Static geometries only (401/160: 16 kHz, 201/80: 8 kHz); everything else keeps the three-launch path.
#pragma once
#include "leaf_fft_wg.hpp"

namespace {

constexpr int kSmallWaves = 11;                  // 10 block waves + the table wave
constexpr int kSmallSplitWaves = 7;              // SPLIT: 6 block waves + the table wave (at most two waves per SIMD)
constexpr int kSmallSplitRing = kSmallSplitWaves - 1;
constexpr int kSmallRing = 10;                   // most block spectra resident at once (one pass = up to this many blocks)
constexpr int kSmallMaxBlocks = 2 * kSmallRing;  // clips up to two passes long take this kernel

struct SmallParams {
    const void* x;          // [B][T] fp32, or bf16 when io_bf16
    int io_bf16;
    const float* kernel;    // [F][2] (mu, sigma), unclamped
    const float* pool_w;    // [F]
    GaborBounds bd;
    int B, T, TP, F, nblk;
    int ring;               // block spectra held in LDS per pass (<= kSmallRing)
    FinParams fin;          // part unused: the sums stay in LDS
    // SPLIT variant (two workgroups per (clip, filter), grid (F, B, 2)): the EMA state the first half hands to the second,
    // [B][F] pairs (epoch, value bits) in the workspace; `epoch` is this launch's 64-bit ticket (host counter from a random seed)
    unsigned long long epoch;
    unsigned long long* carry;
};

constexpr unsigned leaf_layout_hash_small() {
    return leaf_mix(leaf_mix(leaf_mix(leaf_mix(leaf_layout_hash_fft(), sizeof(SmallParams)), offsetof(SmallParams, fin)), offsetof(SmallParams, ring)),
                    offsetof(SmallParams, carry));
}

// dynamic LDS: twiddles | R (the taps first) | full transposition scratch of every wave | frame sums | finalize coefficients |
// pooling-weight vectors.  (Round 5: no spectrum ring -- a wave keeps its block's spectrum in registers between the phases.)
constexpr int kSmallPwFloats = 17 * 64;          // wg_pool_nj <= 17 vectors of 64 lanes (201/80: 17, 401/160: 15)
inline size_t fft_small_lds_bytes(int waves, int TP) {
    return ((size_t)kTwFloats + kFftN + (size_t)waves * kWgScrFloats + (size_t)((TP + 3) / 4 * 4) + 8 + kSmallPwFloats) * 4;
}
inline size_t fft_small_split_lds_bytes(int TP) { return fft_small_lds_bytes(kSmallSplitWaves, TP); }
// SPLIT: the first half takes blocks 0 .. nblk / 2 - 1 and every frame whose window ends inside them; the second half the other
// frames and the blocks their windows meet (one block is transformed by both halves)
__host__ __device__ inline int fft_small_split_frame(int nblk, int LS, int K, int padL, int hop, int TP) {
    const int ms = ((nblk / 2) * LS - K + padL) / hop + 1;
    return ms < 0 ? 0 : ms > TP ? TP : ms;
}

// Inclusive scan of affine maps M -> a M + b down the 64 lanes (lane 0's map applied first): afterwards lane k holds the
// composition of lanes 0..k.  Six DPP steps (row_shr 1, 2, 4, 8 inside the rows of 16, then row_bcast:15 / :31 across them)
// instead of a 64-step dependent chain: the EMA recurrence M_m = w p_m + (1 - w) M_{m-1} (postprocessing.py:22) of a whole
// chunk of frames in ~40 instructions.  The association order differs from the sequential loop's: the smoothed value agrees with
// fin_ema_step's to ~1e-7 relative (every factor is in (0, 1]; tested against the oracle and the row kernel), not to the bit.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ void affine_scan_step(float& a, float& b) {
#pragma clang fp contract(off)
    const float ap = __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(1.0f), __float_as_int(a), CTRL, ROW_MASK, 0xf, false));
    const float bp = __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(0.0f), __float_as_int(b), CTRL, ROW_MASK, 0xf, false));
    b = a * bp + b;                                                       // this lane's map after the predecessor's: a (ap M + bp) + b
    a = a * ap;
}
__device__ __forceinline__ void wave_affine_scan(float& a, float& b) {
    affine_scan_step<0x111, 0xf>(a, b);                                   // row_shr:1
    affine_scan_step<0x112, 0xf>(a, b);                                   // row_shr:2
    affine_scan_step<0x114, 0xf>(a, b);                                   // row_shr:4
    affine_scan_step<0x118, 0xf>(a, b);                                   // row_shr:8
    affine_scan_step<0x142, 0xa>(a, b);                                   // row_bcast:15 -> rows 1, 3
    affine_scan_step<0x143, 0xc>(a, b);                                   // row_bcast:31 -> rows 2, 3
}

// SPLIT (round 5): two workgroups per (clip, filter), grid (F, B, 2), seven waves each -- at most two waves per SIMD, where a wave
// issues every ~5 cycles instead of every ~8.7 at three (profiles/r04/ubench_valu.txt), and the full transposition scratch.  The
// halves split the FRAMES of the row (fft_small_split_frame) and each transforms the blocks its frames' windows meet; the EMA
// state at the seam travels through the workspace: the first half publishes it (value, then this launch's 64-bit ticket with
// release), the second half's finalize wave scans its frames' recurrence first and waits for the ticket only to apply the state.  First halves have the lower workgroup ids: they are dispatched
// first and wait for nothing, so the wait always ends.
template <int SK, int SHOP, bool SPLIT = false>
__global__ __launch_bounds__((SPLIT ? kSmallSplitWaves : kSmallWaves) * 64, SPLIT ? 2 : 3) void leaf_fft_small_kernel(const SmallParams p) {
    constexpr int NW = SPLIT ? kSmallSplitWaves : kSmallWaves;
    constexpr int SCRF = kWgScrFloats;
    constexpr int PADL = SK / 2 + SK % 2 - 1;
    constexpr int LS = fft_block_len(SK, SHOP, true);
    constexpr int DMIN = -((SK - 1 - PADL) / SHOP);
    constexpr int DMAX = (LS - 1 + PADL) / SHOP;
    constexpr int NFR = DMAX - DMIN + 1;
    constexpr int NROW = LS / 64;
    constexpr int NGRP = (NFR + 15) / 16;
    constexpr int PG = wg_pool_step(SHOP), PJ0 = wg_pool_jmin(SK, SHOP), NJ = wg_pool_nj(SK, SHOP);
    static_assert(LS % SHOP == 0 && LS % 64 == 0 && LS > 0 && NFR <= 32 && (SK & 1) && SK <= kFftN / 2 + 1, "static odd-window geometry");
    static_assert((PADL - PJ0) % PG == 0, "window offsets are congruent to padL modulo gcd(64, hop)");

    extern __shared__ __attribute__((aligned(16))) float ssm[];
    float2* twl = reinterpret_cast<float2*>(ssm);                        // [32][64]
    float2* twp = twl + 32 * 64;                                          // [2][16][2]
    float* R = reinterpret_cast<float*>(twp + 64);                        // [2048]; first the taps, conj(w)[K] as float2
    float* scr0 = R + kFftN;
    float* lsum = scr0 + (size_t)NW * SCRF;                               // [TP]
    FinCoef* cfs = reinterpret_cast<FinCoef*>(lsum + (p.TP + 3) / 4 * 4);  // the row's finalize coefficients (phase 0 -> phase 3)
    float* gw = reinterpret_cast<float*>(cfs + 1);                        // [NJ][64]: the filter's pooling-weight vectors (phase 0 -> phase 2)
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    int lane = tid & 63;
    const int f = blockIdx.x, b = blockIdx.y;
    float* scr = scr0 + (size_t)wave * SCRF;
    // SPLIT: this half's frames [m_a, m_b) and blocks [c_first, c_first + nb_half)
    int c_first = 0, nb_half = p.nblk, m_a = 0, m_b = p.TP;
    const int half = SPLIT ? (int)blockIdx.z : 0;
    if constexpr (SPLIT) {
        const int ms = fft_small_split_frame(p.nblk, LS, SK, PADL, SHOP, p.TP);
        if (half == 0) { m_b = ms; nb_half = p.nblk / 2; }
        else { m_a = ms; c_first = max(0, ms * SHOP - PADL) / LS; nb_half = p.nblk - c_first; }
    }
    const unsigned scr_lds = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(__attribute__((address_space(3))) float*)scr);

#ifdef LEAF_SMALL_STAMP             // experiment builds (tools/small_stamps.py): s_memtime at the phase boundaries of wave 0, into out[0..]
    unsigned long long stamp[8];
    int nstamp = 0;
#define SMALL_STAMP() do { if (nstamp < 8) stamp[nstamp++] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define SMALL_STAMP() do { } while (0)
#endif
    SMALL_STAMP();
    // ---- phase 0.  The filter's parameters (uniform: scalar loads) are requested first: their latency runs under the twiddle tables.
    const float mu = p.kernel[2 * f], sg = p.kernel[2 * f + 1], pw_raw = p.pool_w[f];
    const float* xb = static_cast<const float*>(p.x) + (size_t)b * p.T;
    const unsigned short* xh = static_cast<const unsigned short*>(p.x) + (size_t)b * p.T;
    float zre[32], zim[32];             // a block wave's samples -> spectrum (kept across the barrier) -> filter outputs
    auto load_block = [&](int c, int lane_) {                             // block c, rotated left by padL samples
        const int n_c = c * LS;
        if (p.io_bf16) {
#pragma unroll
            for (int r = 0; r < 32; ++r) {
                const int i = 64 * r + lane_;
                const int n = n_c - PADL + ((i + PADL) & (kFftN - 1));
                const unsigned v = xh[min(max(n, 0), p.T - 1)];
                zre[r] = (n >= 0 && n < p.T) ? __uint_as_float(v << 16) : 0.0f;
                zim[r] = 0.0f;
            }
        } else {
#pragma unroll
            for (int r = 0; r < 32; ++r) {
                const int i = 64 * r + lane_;
                const int n = n_c - PADL + ((i + PADL) & (kFftN - 1));
                zre[r] = (n >= 0 && n < p.T) ? xb[n] : 0.0f;
                zim[r] = 0.0f;
            }
        }
    };
#pragma unroll
    for (int r = 0; r < 32; ++r) { zre[r] = 0.0f; zim[r] = 0.0f; }
    // (requesting the first pass's samples here, under the table build, was measured twice -- round 4 and round 5 -- and lost both
    // times: 32 loads with their index arithmetic ahead of the tables cost phase 0 more (3.9 k -> 6.0 k cycles) than phase 1 won)
    fft_build_twiddles_wg(twl, twp, tid, NW * 64);
    {
        float2* taps = reinterpret_cast<float2*>(R);
        for (int j = tid; j < SK; j += NW * 64) {
            float a, c;
            gabor_tap(mu, sg, p.bd, (float)(j - SK / 2), a, c);
            taps[j] = make_float2(a, -c);                                 // conj(w), as fft_prep_kernel
        }
    }
    for (int m = tid; m < p.TP; m += NW * 64) lsum[m] = 0.0f;
    if (tid == 64) cfs[0] = fin_coef(p.fin, f);                           // (its parameter loads land under the transforms)
    {
        // the pooling weights of this filter, NJ vectors per lane (wg_pool_nj): w_k[lane] = g_f[PJ0 + PG k + lane], zero outside
        // the window -- the values fft_prep_kernel writes into its table row (impulse_responses.py:74-80)
        const float half = 0.5f * (float)(SK - 1);
        const float den = pool_sigma(pw_raw, SK) * half;
        for (int t = tid; t < NJ * 64; t += NW * 64) {
            const int j = PJ0 + PG * (t >> 6) + (t & 63);
            const float q = ((float)j - half) / den;
            const float v = expf(-0.5f * (q * q));
            gw[t] = (j >= 0 && j < SK) ? v : 0.0f;
        }
    }
    __syncthreads();
    SMALL_STAMP();

    // Phases 1 and 2 of every pass run through ONE copy of the wave-level transform: the kernel's code is executed once
    // per launch on every CU, from a cold instruction cache (52 KB with three inlined transforms measured ~17 us per launch, most
    // of it instruction fetch) -- so the loop below is deliberately NOT unrolled: step 2 r = the forward transforms of pass r
    // (and, at r = 0, the table wave's), step 2 r + 1 = its filter tasks, and the second trip through fft2048w finds it cached.
    const int nsteps = SPLIT ? 2 : 2 * ((p.nblk + p.ring - 1) / p.ring);
#pragma nounroll
    for (int step = 0; step < nsteps; ++step) {
        const bool inv = (step & 1) != 0;
        const int c0 = SPLIT ? c_first : (step >> 1) * p.ring;
        const int nb = SPLIT ? nb_half : min(p.ring, p.nblk - c0);        // blocks of this pass
        const bool table = !inv && step == 0 && wave == NW - 1;
        if (wave < nb || table) {
            asm volatile("" : "+v"(lane));
            const int c = c0 + wave, n_c = c * LS;
            if (table) {
                // the filter's spectrum: taps in zero-phase layout (tap j at index (j - K/2) mod N), so that the Hermitian symmetry
                // about the centre tap makes it real; the blocks are loaded rotated to match (fft_prep_kernel, real_spec)
                const float2* taps = reinterpret_cast<const float2*>(R);
#pragma unroll
                for (int r = 0; r < 32; ++r) {
                    const int i = 64 * r + lane;
                    const int j = (i < kFftN / 2 ? i : i - kFftN) + SK / 2;
                    const float2 t = taps[min(max(j, 0), SK - 1)];
                    zre[r] = (j >= 0 && j < SK) ? t.x : 0.0f;
                    zim[r] = (j >= 0 && j < SK) ? t.y : 0.0f;
                }
            } else if (!inv) {
                load_block(c, lane);
            } else {
                // Z = conj(A') R_f, in place: the block's spectrum is in this wave's registers, register i <-> bin 64 brev5(i) + lane
                // (every bin: the block is real, but nothing is mirrored here), and the next transform takes element 64 k + lane in
                // register k -- the value that sits in register brev5(k).  Pairs (k, brev5(k)) swap, palindromes stay.
                float rq[32];                                             // R_f[64 k + lane]
#pragma unroll
                for (int k = 0; k < 32; ++k) rq[k] = R[64 * k + lane];
#pragma unroll
                for (int k = 0; k < 32; ++k) {
                    const int j = brev5(k);
                    if (j == k) {
                        zre[k] = zre[k] * rq[k];
                        zim[k] = -(zim[k] * rq[k]);
                    } else if (j > k) {
                        const float ar = zre[k], ai = zim[k];
                        zre[k] = zre[j] * rq[k];
                        zim[k] = -(zim[j] * rq[k]);
                        zre[j] = ar * rq[j];
                        zim[j] = -(ai * rq[j]);
                    }
                }
            }
            pin32(zre);
            pin32(zim);
            fft2048w<false>(zre, zim, scr, scr_lds, twl, twp, lane);     // register i <-> element 64 brev5(i) + lane
            pin32(zre);
            pin32(zim);
            if (table) {
#pragma unroll
                for (int i = 0; i < 32; ++i) R[64 * brev5(i) + lane] = zre[i] * (1.0f / kFftN);   // imaginary parts: rounding noise
            } else if (inv) {
                const int Lv = min(LS, p.T - n_c);
                int mlo = n_c + PADL - SK + 1;                            // first frame whose window reaches the block
                mlo = mlo <= 0 ? 0 : (mlo + SHOP - 1) / SHOP;
                const int mhi = min(p.TP - 1, (n_c + Lv - 1 + PADL) / SHOP);
                float er[NROW];
#pragma unroll
                for (int i = 0; i < 32; ++i) {
                    const int r = brev5(i);
                    if (r < NROW) er[r] = zre[i] * zre[i] + zim[i] * zim[i];
                }
                if (Lv < LS) {                                            // a clip's last block: outputs past the clip's end
#pragma unroll
                    for (int r = 0; r < NROW; ++r) er[r] = 64 * r + lane < Lv ? er[r] : 0.0f;
                }
                float pw[NJ];                                             // the filter's weight vectors (phase 0)
#pragma unroll
                for (int k = 0; k < NJ; ++k) pw[k] = gw[64 * k + lane];
                float acc[NGRP][16];
#pragma unroll
                for (int g = 0; g < NGRP; ++g)
#pragma unroll
                    for (int fi = 0; fi < 16; ++fi) acc[g][fi] = 0.0f;
#pragma unroll
                for (int r = 0; r < NROW; ++r) {
#pragma unroll
                    for (int fi = 0; fi < NFR; ++fi) {
                        const int is = (DMIN + fi) * SHOP - PADL;
                        if (is <= 64 * r + 63 && is + SK > 64 * r)
                            acc[fi / 16][fi % 16] = fmaf(er[r], pw[(64 * r - is - PJ0) / PG], acc[fi / 16][fi % 16]);
                    }
                }
                asm volatile("" : "+v"(acc[0][0]));
#pragma unroll
                for (int g = 0; g < NGRP; ++g) {
                    const float v = frame_butterfly16(acc[g], lane);
                    const int fi = 16 * g + ((lane >> 5) & 1) * 8 + ((lane >> 4) & 1) * 4 + ((lane >> 3) & 1) * 2 + ((lane >> 2) & 1);
                    const int m = n_c / SHOP + DMIN + fi;
                    if ((lane & 3) == 0 && fi < NFR && m >= mlo && m <= mhi && (!SPLIT || (m >= m_a && m < m_b)))
                        __hip_atomic_fetch_add(&lsum[m], v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                }
            }
        }
        __syncthreads();            // R complete / sums complete
        SMALL_STAMP();
    }
    // ---- phase 3: the row (b, f), by ONE wave -- the sums are in LDS already added up (exactly what the workgroup kernel's tail
    // reads), and one row is T' frames of latency, not throughput: the tile machinery of fft_finalize_tile (704 threads, three
    // staged passes, a barrier each) measured 3.4 us here; lane = frame, 64 at a time: pooled value and floor lane-parallel,
    // the EMA recurrence as a lane scan (round 4: a 64-step DPP chain, bit-identical to the sequential loop, ~1.5 k cycles per
    // chunk), the PCEN point function lane-parallel.
    if (wave == 0) {
        const FinParams& fin = p.fin;
        const int row = b * p.F + f, mode = fin.mode;
        const FinCoef cf = cfs[0];
        const bool scaled = fin.clip_scale2 != nullptr;
        const float s2 = scaled ? fin.clip_scale2[b] : 1.0f;
        float M = 0.0f;
        // The row is scanned in the SAME chunks by both variants of the kernel -- 64 frames at a time from frame 0 and again from
        // the seam frame of the SPLIT form (fft_small_split_frame; clips of 2..10 blocks, whether or not this launch is split) -- so
        // that a clip's bits do not depend on which variant its batch size selects.
        const int seam = (p.nblk >= 2 && p.nblk <= 2 * (kSmallSplitRing - 1)) ? fft_small_split_frame(p.nblk, LS, SK, PADL, SHOP, p.TP) : p.TP;
        for (int seg = 0; seg < (SPLIT ? 1 : 2); ++seg)
        for (int m0 = SPLIT ? m_a : seg ? seam : 0, m_e = SPLIT ? m_b : seg ? p.TP : seam; m0 < m_e; m0 += 64) {
            const int m = m0 + lane;
            const bool on = m < m_e;
            float x = fin_pooled(lsum[on ? m : 0], 0.0f, 0.0f, 1, scaled, s2, cf.bias);
            if (on && fin.raw_out) fin.raw_out[(size_t)row * p.TP + m] = x;
            if (!(mode & 8)) x = pooled_floor(x);
            // The recurrence of the chunk as a lane scan of affine maps (wave_affine_scan): lane k ends up with (A_k, B_k),
            // M_k = A_k carry + B_k, carry = the state before the chunk's first frame -- M_{-1} = p_0 for the clip's first chunk
            // (postprocessing.py:15), the previous chunk's last frame otherwise, and for the second half of a SPLIT pair the first
            // half's last frame, which is only waited for here: the scan itself ran while the first half was still finalizing.
            float Mv = 0.0f;
            if (mode & 1) {
#pragma clang fp contract(off)
                float sa = on ? cf.omw : 1.0f, sb = on ? cf.w * x : 0.0f;
                wave_affine_scan(sa, sb);
                if (SPLIT && half == 1 && m0 == m_a) {
                    // the first half's state after frame m_a - 1: its ticket is this launch's once the value is there
                    unsigned long long* slot = p.carry + 2 * (size_t)row;
                    // (bounded -- ADVICE r5: HIP does not promise that the first halves are dispatched before the second ones.  All
                    // 2 B F workgroups of a SPLIT launch fit the chip at once (the host takes SPLIT only then), so the wait ends as soon
                    // as the first half has a CU; if it does not within ~2^22 polls (seconds: a first half that never ran, or a workspace
                    // shared with another call in flight -- include/leaf_hip.h forbids that) the kernel traps and the stream reports
                    // a launch failure instead of hanging)
                    unsigned polls = 0;
                    while (__hip_atomic_load(slot, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) != p.epoch) {
                        __builtin_amdgcn_s_sleep(8);
                        if (++polls == (1u << 22)) __builtin_trap();
                    }
                    M = __uint_as_float((unsigned)__hip_atomic_load(slot + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
                    // consumed: the ticket is taken out again, so that a REPLAY of this launch (a captured HIP graph carries the
                    // same ticket every time) waits for its own first half instead of reading the previous replay's state
                    if (lane == 0) __hip_atomic_store(slot, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
                const float carry = m0 == 0 ? __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(x))) : M;
                Mv = sa * carry + sb;
                const int n = min(64, m_e - m0);
                M = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(Mv), n - 1));      // the chunk's last frame: the next chunk's carry
                if (SPLIT && half == 0 && m0 + 64 >= m_e && lane == 0) {                        // the seam: out before this chunk's point functions
                    unsigned long long* slot = p.carry + 2 * (size_t)row;
                    __hip_atomic_store(slot + 1, (unsigned long long)__float_as_uint(M), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    __hip_atomic_store(slot, p.epoch, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
                }
            }
            const float o = fin_point(cf, mode, fin.floor_, x, Mv);
            if (on) fin_store(fin, (size_t)row * p.TP + m, o);
        }
    }
#ifdef LEAF_SMALL_STAMP
    SMALL_STAMP();
    __syncthreads();
    if (blockIdx.x == 0 && blockIdx.y == 0 && tid == 0)                   // (SPLIT: the second half's stamps behind the first's)
        for (int i = 1; i < nstamp; ++i) static_cast<float*>(p.fin.out)[8 * half + i - 1] = (float)(stamp[i] - stamp[0]);
#endif
#undef SMALL_STAMP
}

}  // namespace
