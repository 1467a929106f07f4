// leaf_fft_wg_bwd.hpp -- overlap-save BACKWARD on the workgroup-per-block structure of leaf_fft_wg.hpp, with dL/dx
// Part of the single translation unit leaf_kernels.hip (gfx950 only); see that file's header comment.
//
// Same task queue, same LDS ring for the block spectrum A' as the forward kernel; an inverse task (block, filter f) is the
// backward epilogue of leaf_fft_kernel<.., BWD = 1> (leaf_fft.hpp), i.e. with u = conj(y) = FFT(conj(A' R_f)) in registers:
//   transposed pooling  de[n] = sum_m g_pre[m] g_f[n - m hop + padL]   (+ this block's share of d pool_w),
//   gy = 2 de y  ->  second transform  g = FFT(gy) = dL/dS  (S = A' R_f),
//   dL/dR[k] = Re(conj(A'[k]) g[k])  ->  d mu, d sigma as two spectral dot products with R_mu, R_sigma,
// written as per-(block, filter) partials (deterministic, no atomics).
//
// dL/dx (what autograd yields through convolution.py:97): dL/dA'[k] = sum_f R_f[k] g_f[k] =: G[k], and because the block a'
// is real, dL/da' = Re(FFT(conj G)): the sum over filters costs no transform, the block needs ONE more.  The sum must
// live somewhere while the block's filters are processed, and the filters of a block are spread over twelve waves.
// Accumulating it in LDS with ds_add_f32 measured 2.4 ms per backward (the LDS serialises float atomics lane by lane).
// Round 2 therefore gave every WAVE a whole block (leaf_fft_blk_bwd_dx_kernel: the spectrum A' in wave-private LDS, G in 64
// registers across the block's filter loop at the 256-VGPR cap, two waves per SIMD) -- still what runs below one block per
// CU and at K = 801.  Round 3 keeps the twelve-wave structure (leaf_fft_wg_bwd_kernel<.., DX = true>): G per block in LDS,
// Hermitian-folded, added to by plain read-add-write in FILTER ORDER through a ticket (wg_dx_accumulate / wg_dx_finish
// below) -- no float atomics, and the sum order does not depend on timing.  Either way the block's 2048 input-gradient
// samples go, un-rotated, into dxblk[block][2048]; fft_dx_gather_kernel sums the (at most three) overlapping blocks of every
// sample in a fixed order.  No atomics on data anywhere: bit-reproducible.
#pragma once
#include "leaf_fft_wg.hpp"
#ifndef LEAF_DX_PRIO
#define LEAF_DX_PRIO 1                 // the wave whose turn it is goes first on its SIMD until it has passed the ticket on (0: A/B)
#endif
#include "leaf_band_bwd.hpp"

// Static backward kernels: the filter's pooling weights as NJ register vectors per lane (wg_pool_nj: 13 at 401 / 160 -- the
// forward's form since round 3) instead of a wave-private LDS row filled by DMA and ~80 ds_read_b32 per task.  Same-box A/B,
// gradients bit-identical (profiles/r04/ab_bwd_regw.txt): whole backward 0.4657 -> 0.4466 ms at cfg1, -2.5 % with dL/dx,
// -2.8 % at 8 kHz, -4.0 % at 512 clips.  0: the LDS row (A/B).
#ifndef LEAF_WG_BWD_REGW
#define LEAF_WG_BWD_REGW 1
#endif
#ifndef LEAF_BAND_BWD
#define LEAF_BAND_BWD 1                // the static 401 / 160 backward (parameter gradients) runs the narrow-band filters as band tasks (leaf_band_bwd.hpp); 0: A/B
#endif
#ifndef LEAF_BAND_BWD_DX
#define LEAF_BAND_BWD_DX 1             // ... and with dL/dx: the band tasks add their members' shares of the block's gradient spectrum (DXB); 0: A/B
#endif
#ifndef LEAF_WG_BWD_FUSE2
#define LEAF_WG_BWD_FUSE2 (LEAF_FFT32_DIT && LEAF_FFT_FUSE_TWIDDLE)    // gy's rows in pairs (r, r + 16) with the second transform's first stage; 0: A/B
#endif
#ifndef LEAF_WG_BWD_PW2
#define LEAF_WG_BWD_PW2 1              // ... and a second set, the weights times (tap - centre)^2 (d pool_w); 0: squared per use (A/B)
#endif

namespace {

// entries k = 0..15 of a 16 x 64 float2 LDS table read with inline-asm ds_read_b64 (see lds_rd8): body(idx, value)
template <typename OffFn, typename Body>
__device__ __forceinline__ void lds_stream16(unsigned addr, OffFn, Body body) {
    v2f buf[2][8];
    lds_rd8<OffFn::off(0)>(buf[0][0], addr); lds_rd8<OffFn::off(1)>(buf[0][1], addr); lds_rd8<OffFn::off(2)>(buf[0][2], addr);
    lds_rd8<OffFn::off(3)>(buf[0][3], addr); lds_rd8<OffFn::off(4)>(buf[0][4], addr); lds_rd8<OffFn::off(5)>(buf[0][5], addr);
    lds_rd8<OffFn::off(6)>(buf[0][6], addr); lds_rd8<OffFn::off(7)>(buf[0][7], addr);
    lds_rd8<OffFn::off(8)>(buf[1][0], addr); lds_rd8<OffFn::off(9)>(buf[1][1], addr); lds_rd8<OffFn::off(10)>(buf[1][2], addr);
    lds_rd8<OffFn::off(11)>(buf[1][3], addr); lds_rd8<OffFn::off(12)>(buf[1][4], addr); lds_rd8<OffFn::off(13)>(buf[1][5], addr);
    lds_rd8<OffFn::off(14)>(buf[1][6], addr); lds_rd8<OffFn::off(15)>(buf[1][7], addr);
    lds_wait8<8>(buf[0]);
#pragma unroll
    for (int j = 0; j < 8; ++j) body(j, buf[0][j]);
    lds_wait8<0>(buf[1]);
#pragma unroll
    for (int j = 0; j < 8; ++j) body(8 + j, buf[1][j]);
}
// The block spectrum as stored by the forward task (bins 0..1024): body(k, re, im) with (re, im) = A'[64 k + lane],
// k = 0..31 -- rows 0..15 straight, rows 16..31 as the conjugate of the mirrored bin (A'[N - e] = conj(A'[e])).
template <typename Body>
__device__ __forceinline__ void wg_ring_rows(const float2* A, int lane, Body body) {
    lds_stream16(lds_addr(A + lane), OffRow{}, [&](int k, v2f a) { body(k, a.x, a.y); });
    lds_stream16(lds_addr(A + (kFftN - 64 * 31) - lane), OffRow{}, [&](int j, v2f m) { body(31 - j, m.x, -m.y); });
}

// Spectral tail of one (block, filter): g = dL/dS in (vre, vim) (register brev5(k) <-> bin 64 k + lane), A' in LDS at A.
// dL/dR[k] = Re(conj(A'[k]) g[k]); d mu, d sigma = <dL/dR, R_mu>, <dL/dR, R_sigma> are ADDED to this lane's (amu, asg)
// (before the wave sums); DX: G += R_f g into (acc_re, acc_im).
template <int DX>
__device__ __forceinline__ void wg_bwd_tail(const FftParams& p, const float2* A, int lane, int f, const float (&vre)[32],
                                            const float (&vim)[32], float (&acc_re)[32], float (&acc_im)[32], float& amu,
                                            float& asg) {
    {
        const float* rmu = reinterpret_cast<const float*>(p.H) + ((size_t)p.F + f) * kFftN + lane;
        const float* rsg = reinterpret_cast<const float*>(p.H) + ((size_t)2 * p.F + f) * kFftN + lane;
        // (vre[brev5(k)], vim[brev5(k)]) = g[64 k + lane]: the transform leaves its output bit-reversed over registers
        const float* rr = reinterpret_cast<const float*>(p.H) + (size_t)f * kFftN + lane;   // R_f again (DX): cheaper than
        auto chunk = [&](auto cc) {                                    // carrying 32 registers through both transforms
            constexpr int C = decltype(cc)::value;                     // rows 8 C .. 8 C + 7 (C < 2), mirrored rows for C >= 2
            float tm[8], ts[8], tr8[8];
            // the previous chunk's sums are complete before this chunk's loads issue (otherwise all four chunks' loads
            // are hoisted to the front, their 128 destination registers do not fit, and each one is spilled)
            asm volatile("" : "+v"(amu), "+v"(asg) : : "memory");
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int k = C < 2 ? 8 * C + j : 31 - (8 * (C - 2) + j);
                tm[j] = rmu[64 * k];
                ts[j] = rsg[64 * k];
                tr8[j] = DX ? rr[64 * k] : 0.0f;
            }
            asm volatile("" ::: "memory");
            v2f a[8];
            const unsigned base = C < 2 ? lds_addr(A + lane) : lds_addr(A + (kFftN - 64 * 31) - lane);
            lds_rd8<512 * (8 * (C & 1) + 0)>(a[0], base); lds_rd8<512 * (8 * (C & 1) + 1)>(a[1], base);
            lds_rd8<512 * (8 * (C & 1) + 2)>(a[2], base); lds_rd8<512 * (8 * (C & 1) + 3)>(a[3], base);
            lds_rd8<512 * (8 * (C & 1) + 4)>(a[4], base); lds_rd8<512 * (8 * (C & 1) + 5)>(a[5], base);
            lds_rd8<512 * (8 * (C & 1) + 6)>(a[6], base); lds_rd8<512 * (8 * (C & 1) + 7)>(a[7], base);
            lds_wait8<0>(a);
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int k = C < 2 ? 8 * C + j : 31 - (8 * (C - 2) + j);
                const float ar = a[j].x, ai = C < 2 ? a[j].y : -a[j].y;           // A'[64 k + lane]
                const float gr = vre[brev5(k)], gi = vim[brev5(k)];
                const float d = ar * gr + ai * gi;                                // dL/dR[k]
                amu = fmaf(d, tm[j], amu);
                asg = fmaf(d, ts[j], asg);
                if constexpr (DX) {                                               // G += R_f g at bin 64 k + lane
                    acc_re[k] = fmaf(tr8[j], gr, acc_re[k]);
                    acc_im[k] = fmaf(tr8[j], gi, acc_im[k]);
                }
            }
        };
        chunk(std::integral_constant<int, 0>{});
        chunk(std::integral_constant<int, 1>{});
        chunk(std::integral_constant<int, 2>{});
        chunk(std::integral_constant<int, 3>{});
    }
}

// ---- dL/dx on the workgroup structure (leaf_fft_wg_bwd_kernel / leaf_fft_wgg_bwd_kernel with DX = true).
// dL/dA'[k] = sum_f R_f[k] g_f[k] =: G[k] is accumulated per BLOCK in LDS, one array per ring slot, by the waves that run the
// block's filters; the wave that adds the last filter turns it into the block's 2048 input-gradient samples.  The array is
// Hermitian-folded: dL/da' = Re(FFT(conj G)) only sees G[k] + conj(G[N - k]), so bins k > 1024 are added, conjugated, at
// N - k and the array has 1025 entries.  The filters of a block add in filter order (ticket = filters added so far: the queue
// hands the filters out in that order, so the predecessor is always running) -- plain read-add-write, no float atomics,
// and the sum order does not depend on timing: bit-reproducible like everything else.
//
// wg_dx_accumulate: (vre, vim) = g_f, register brev5(k) <-> bin 64 k + lane (destroyed).
__device__ __forceinline__ void wg_dx_accumulate(const FftParams& p, int f, int lane, float (&vre)[32], float (&vim)[32], float2* gS,
                                                 const int* gticket, int want) {
    // the products R_f g_f first, in place, eight table values at a time; then the turn
    {
        const float* rr = reinterpret_cast<const float*>(p.H) + (size_t)f * kFftN + lane;
#pragma unroll
        for (int k0 = 0; k0 < 32; k0 += 8) {
            float rv[8];
            // one chunk's loads at a time (registers): the chunk's POINTER is made opaque, in the global address space, so that
            // its eight loads differ by an immediate offset (an opaque index costs ~3 VALU per load in 64-bit address arithmetic)
            const __attribute__((address_space(1))) float* q = (const __attribute__((address_space(1))) float*)rr + 64 * k0;
            asm volatile("" : "+v"(q) : : "memory");
#pragma unroll
            for (int j = 0; j < 8; ++j) rv[j] = q[64 * j];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                vre[brev5(k0 + j)] *= rv[j];
                vim[brev5(k0 + j)] *= rv[j];
            }
        }
    }
#ifndef LEAF_DX_NOWAIT                 // measurement only (wrong sums): what the ordered turn costs
    wg_wait_ge(gticket, want);
#endif
    if (LEAF_DX_PRIO) __builtin_amdgcn_s_setprio(3);                      // (the turns are one dependent chain through the block)
    // (eight reads in flight per step; sixteen measured slower: 22.05 kHz 1.93 -> 2.01 ms, 48 kHz 5.59 -> 5.97 ms with dL/dx)
    float2* s1 = gS + lane;                                               // bin 64 k + lane, k < 16
#pragma unroll
    for (int k0 = 0; k0 < 16; k0 += 8) {
        float2 sv[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) sv[j] = s1[64 * (k0 + j)];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int k = k0 + j;
            sv[j].x += vre[brev5(k)];
            sv[j].y += vim[brev5(k)];
            s1[64 * k] = sv[j];
        }
    }
    float2* s2 = gS + (kFftN - 64 * 31) - lane;                          // bin k' = 64 k + lane >= 1024 lands, conjugated, at N - k'
#pragma unroll
    for (int k0 = 16; k0 < 32; k0 += 8) {
        float2 sv[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) sv[j] = s2[64 * (31 - (k0 + j))];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int k = k0 + j;
            sv[j].x += vre[brev5(k)];
            sv[j].y -= vim[brev5(k)];
            s2[64 * (31 - k)] = sv[j];
        }
    }
    wg_release();
    if (lane == 0) __hip_atomic_fetch_add(const_cast<int*>(gticket), 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    if (LEAF_DX_PRIO) __builtin_amdgcn_s_setprio(0);
}
// wg_dx_finish: called by the wave that added the block's last filter (filters add in order, so every other one is in).
// X = the Hermitian spectrum whose transform is dL/da': X[k] = conj(S[k]) / 2 (0 < k < 1024), X[N - k] = S[k] / 2,
// X[0] = Re S[0], X[1024] = Re S[1024] (S[0] holds G[0], S[1024] conj(G[1024]): each was written by one of the two passes
// only).  Sample i of the rotated block is x[n_c - padL + ((i + rot) mod N)]: stored un-rotated into part[gb][2048].
// tS: the unpaired tap's time-domain sums of an even window (2048 floats, un-rotated block coordinates), or nullptr.
template <bool HALF>
__device__ __forceinline__ void wg_dx_finish(const FftParams& p, const float2* gS, const float* tS, int gb, int rot, int lane, float* scr,
                                             unsigned scr_lds, const float2* twl, const float2* twh) {
    float xre[32], xim[32];
    const float2* s1 = gS + lane;
    const float2* s2 = gS + (kFftN - 64 * 31) - lane;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const float2 v = s1[64 * r];
        const bool self = r == 0 && lane == 0;
        xre[r] = self ? v.x : 0.5f * v.x;
        xim[r] = self ? 0.0f : -0.5f * v.y;
    }
#pragma unroll
    for (int r = 16; r < 32; ++r) {
        const float2 v = s2[64 * (31 - r)];
        const bool self = r == 16 && lane == 0;
        xre[r] = self ? v.x : 0.5f * v.x;
        xim[r] = self ? 0.0f : 0.5f * v.y;
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    fft2048w<HALF>(xre, xim, scr, scr_lds, twl, twh, lane);
    float* dst = p.part + (size_t)gb * kFftN;
#pragma unroll
    for (int i = 0; i < 32; ++i) {
        const int n = (64 * brev5(i) + lane + rot) & (kFftN - 1);
        dst[n] = tS ? xre[i] + tS[n] : xre[i];
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                    // the slot's arrays are read before the slot can be released
}

// The recomputed forward of a backward task as the forward kernel does it (leaf_fft_wg_kernel): the spectral multiply
// Z = conj(A' R_f) fused with the first decimation-in-time stage of the transform that follows (rows (k, k + 16), unit
// twiddles): out[k] = za + zb, out[k + 16] = za - zb as one product and two FMAs per component -- 6 instructions per pair of
// rows instead of 4 products + 4 additions; the transform is then called with SKIP1.  0: separate multiply (A/B).
#ifndef LEAF_WG_BWD_FUSE1
#define LEAF_WG_BWD_FUSE1 (LEAF_FFT32_DIT && LEAF_FFT_FUSE_TWIDDLE)
#endif
__device__ __forceinline__ void wg_multiply_stage1(const float2* A, int lane, const float (&rq)[32], float (&zre)[32], float (&zim)[32]) {
    // two streams of 16 rows: ascending from A[lane], and the mirror A[2048 - 64 k - lane], k = 16..31, read as rows 15..0 of the
    // base A[64 - lane]: row k + 16 is hi[k]
    const unsigned a_lo = lds_addr(A + lane), a_hi = lds_addr(A + (kFftN - 64 * 31) - lane);
    v2f lo[16], hi[16];
    auto rd = [&](auto kk) {
        constexpr int k = decltype(kk)::value;
        if constexpr (k < 16) lds_rd8<512 * k>(lo[k], a_lo);
        else lds_rd8<512 * (31 - k)>(hi[k - 16], a_hi);
    };
#define LEAF_RD8(B) rd(std::integral_constant<int, B + 0>{}); rd(std::integral_constant<int, B + 1>{}); \
                    rd(std::integral_constant<int, B + 2>{}); rd(std::integral_constant<int, B + 3>{}); \
                    rd(std::integral_constant<int, B + 4>{}); rd(std::integral_constant<int, B + 5>{}); \
                    rd(std::integral_constant<int, B + 6>{}); rd(std::integral_constant<int, B + 7>{});
    v2f(&lo0)[8] = *reinterpret_cast<v2f(*)[8]>(&lo[0]);
    v2f(&lo1)[8] = *reinterpret_cast<v2f(*)[8]>(&lo[8]);
    v2f(&hi0)[8] = *reinterpret_cast<v2f(*)[8]>(&hi[0]);
    v2f(&hi1)[8] = *reinterpret_cast<v2f(*)[8]>(&hi[8]);
    auto pair = [&](int k) {
        const float ra = rq[k], rb = rq[k + 16];
        const float tr_ = lo[k].x * ra, ti_ = -(lo[k].y * ra);
        zre[k] = fmaf(hi[k].x, rb, tr_);
        zim[k] = fmaf(hi[k].y, rb, ti_);
        zre[k + 16] = fmaf(-hi[k].x, rb, tr_);
        zim[k + 16] = fmaf(-hi[k].y, rb, ti_);
    };
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                // the counted waits below see only these reads
    LEAF_RD8(0) LEAF_RD8(16) LEAF_RD8(8)
    lds_wait8<8>(lo0);
    lds_wait8<8>(hi0);
#pragma unroll
    for (int k = 0; k < 8; ++k) pair(k);
    LEAF_RD8(24)
    lds_wait8<0>(lo1);
    lds_wait8<0>(hi1);
#pragma unroll
    for (int k = 8; k < 16; ++k) pair(k);
#undef LEAF_RD8
}

// Backward of ONE filter on ONE block whose spectrum A' (bins 0..1024) sits in LDS at A; rq = R_f[64 k + lane].
// Returns this lane's shares of d mu, d sigma and d pool_w (before the wave sums) and, DX, adds R_f g to (acc_re, acc_im).
template <int SK, int SHOP, int DX, bool HALF = true>
__device__ __forceinline__ void wg_bwd_filter(const FftParams& p, const float2* A, int lane, int f, int b, int c,
                                              const float (&rq)[32], float* scr, unsigned scr_lds, float* sG, const float2* twl,
                                              const float2* twh, float (&acc_re)[32], float (&acc_im)[32], float& amu_out,
                                              float& asg_out, float& dpw_out, [[maybe_unused]] float2* gS = nullptr,
                                              [[maybe_unused]] const int* gticket = nullptr, [[maybe_unused]] int want = 0) {
    constexpr int GU = fft_wg_row_floats(SK);
    constexpr int PADL = SK / 2 + SK % 2 - 1;
    constexpr int LS = fft_block_len(SK, SHOP, true);
    constexpr int DMIN = -((SK - 1 - PADL) / SHOP);
    constexpr int DMAX = (LS - 1 + PADL) / SHOP;
    constexpr int NFR = DMAX - DMIN + 1;
    constexpr int NROW = LS / 64;
    const int n_c = c * LS;
    const int Lv = min(LS, p.T - n_c);
    int mlo = n_c + PADL - SK + 1;
    mlo = mlo <= 0 ? 0 : (mlo + SHOP - 1) / SHOP;
    const int mhi = min(p.TP - 1, (n_c + Lv - 1 + PADL) / SHOP);
    float zre[32], zim[32];
#if LEAF_WG_BWD_FUSE1
    wg_multiply_stage1(A, lane, rq, zre, zim);                        // Z = conj(A' R_f) and the transform's first stage
#else
    wg_ring_rows(A, lane, [&](int k, float ar, float ai) {           // Z = conj(A' R_f), natural row order
        zre[k] = ar * rq[k];
        zim[k] = -(ai * rq[k]);
    });
#endif
#if LEAF_WG_BWD_REGW
    constexpr int PG = wg_pool_step(SHOP), PJ0 = wg_pool_jmin(SK, SHOP), NJ = wg_pool_nj(SK, SHOP);
    float pw[NJ];
    {
        const float* gsrc = p.Gz + (size_t)f * p.GZ + (kGPad + PJ0) + lane;
        asm volatile("" ::: "memory");
#pragma unroll
        for (int k = 0; k < NJ; ++k) pw[k] = gsrc[PG * k];
        asm volatile("" ::: "memory");
    }
    (void)sG; (void)GU;
#else
    {   // pooling row of this filter -> wave-private LDS, lands under the transform
        const float* gsrc = p.Gz + (size_t)f * p.GZ;
#pragma unroll
        for (int i0 = 0; i0 < GU; i0 += 256)
            if (i0 + 256 <= GU || i0 + 4 * lane < GU)
                __builtin_amdgcn_global_load_lds(gsrc + i0 + 4 * lane, (__attribute__((address_space(3))) void*)(sG + i0), 16, 0, 0);
        asm volatile("" ::: "memory");
    }
#endif
    fft2048w<HALF, LEAF_WG_BWD_FUSE1 != 0>(zre, zim, scr, scr_lds, twl, twh, lane);   // u = conj(y): register i <-> samples 64 brev5(i) + lane
    pin32(zre);
    pin32(zim);
    // g_pre of the NFR frames this block meets, as wave-uniform scalars
    float gp[NFR];
    {
        const int fi = lane & 31, m = n_c / SHOP + DMIN + fi;
        const float mine = (fi < NFR && m >= mlo && m <= mhi) ? p.gpre[((size_t)b * p.F + f) * p.TP + m] : 0.0f;
#pragma unroll
        for (int qq = 0; qq < NFR; ++qq) gp[qq] = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(mine), qq));
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                  // the row DMA has landed (gpre loads waited for with it)
    constexpr float HALFW = 0.5f * (float)(SK - 1);
    const float lanef = (float)lane;
    float dpw = 0.0f;
#if LEAF_WG_BWD_REGW && LEAF_WG_BWD_PW2
    // the weights times (window position - centre)^2, position = PJ0 + PG k + lane: d pool_w needs sum g (j - c)^2 e, and a
    // second weight vector per offset turns five instructions per (row, frame) into two FMAs
    float pw2[NJ];
#pragma unroll
    for (int k = 0; k < NJ; ++k) {
        const float tj = (float)(PJ0 + PG * k) - HALFW + lanef;
        pw2[k] = pw[k] * (tj * tj);
    }
#endif
    float vre[32], vim[32];                                           // gy = 2 de y, natural row order
#if LEAF_WG_BWD_REGW && LEAF_WG_BWD_PW2 && LEAF_WG_BWD_FUSE2
    // rows r and r + 16 together, and with them the first decimation-in-time stage of the transform that follows (unit twiddles):
    // out[r] = gy[r] + gy[r + 16], out[r + 16] = gy[r] - gy[r + 16] as one product and two FMAs per component; rows past the
    // block's outputs (r + 16 >= NROW) are zero, so both outputs are gy[r]
    auto row_grad = [&](auto rr, float& s2, float& ur, float& ui) {
        constexpr int r = decltype(rr)::value;
        const int i = brev5(r);
        ur = zre[i];
        ui = zim[i];
        const bool ok = 64 * r + lane < Lv;
        float de = 0.0f, dq = 0.0f;
#pragma unroll
        for (int fi = 0; fi < NFR; ++fi) {
            const int is = (DMIN + fi) * SHOP - PADL;
            if (is <= 64 * r + 63 && is + SK > 64 * r) {
                de = fmaf(gp[fi], pw[(64 * r - is - PJ0) / PG], de);
                dq = fmaf(gp[fi], pw2[(64 * r - is - PJ0) / PG], dq);
            }
        }
        const float e = ok ? ur * ur + ui * ui : 0.0f;
        dpw = fmaf(e, dq, dpw);
        s2 = ok ? 2.0f * de : 0.0f;
    };
    auto row_pair = [&](auto rr) {
        constexpr int r = decltype(rr)::value;
        float are = 0.0f, aim = 0.0f;
        if constexpr (r < NROW) {
            float s2a, ura, uia;
            row_grad(rr, s2a, ura, uia);
            are = s2a * ura;
            aim = -(s2a * uia);
        }
        if constexpr (r + 16 < NROW) {
            float s2b, urb, uib;
            row_grad(std::integral_constant<int, r + 16>{}, s2b, urb, uib);
            vre[r] = fmaf(s2b, urb, are);
            vim[r] = fmaf(-s2b, uib, aim);
            vre[r + 16] = fmaf(-s2b, urb, are);
            vim[r + 16] = fmaf(s2b, uib, aim);
        } else {
            vre[r] = vre[r + 16] = are;
            vim[r] = vim[r + 16] = aim;
        }
    };
#define LEAF_ROW4(B0)                                                                                                    \
    row_pair(std::integral_constant<int, B0 + 0>{}); row_pair(std::integral_constant<int, B0 + 1>{});                    \
    row_pair(std::integral_constant<int, B0 + 2>{}); row_pair(std::integral_constant<int, B0 + 3>{});                    \
    asm volatile("" : "+v"(vre[B0 + 3]), "+v"(vim[B0 + 3]), "+v"(vre[B0 + 19]), "+v"(vim[B0 + 19]), "+v"(dpw));   /* groups stay in program order */
    LEAF_ROW4(0) LEAF_ROW4(4) LEAF_ROW4(8) LEAF_ROW4(12)
#undef LEAF_ROW4
    constexpr bool kSkip1 = true;
#else
    constexpr bool kSkip1 = false;
    int gofs = kGPad + lane;                                          // made opaque per row: keeps the rows in program order
#pragma unroll
    for (int r = 0; r < 32; ++r) {
        const int i = brev5(r);                                       // register holding row r of u
        if (r < NROW) {
            const float ur = zre[i], ui = zim[i];
            const bool ok = 64 * r + lane < Lv;
            float de = 0.0f, dq = 0.0f;
            if (r % 4 == 0) asm volatile("" : "+v"(gofs));            // groups of four rows stay in program order
#pragma unroll
            for (int fi = 0; fi < NFR; ++fi) {
                const int is = (DMIN + fi) * SHOP - PADL;
                if (is <= 64 * r + 63 && is + SK > 64 * r) {
#if LEAF_WG_BWD_REGW
#if LEAF_WG_BWD_PW2
                    de = fmaf(gp[fi], pw[(64 * r - is - PJ0) / PG], de);          // zero outside the window
                    dq = fmaf(gp[fi], pw2[(64 * r - is - PJ0) / PG], dq);         // the same weight times (window position - centre)^2
#else
                    const float gw = gp[fi] * pw[(64 * r - is - PJ0) / PG];      // zero outside the window
                    const float tj = (float)(64 * r - is) - HALFW + lanef;        // window position - centre
                    de += gw;
                    dq = fmaf(gw, tj * tj, dq);
#endif
#else
                    const float gw = gp[fi] * sG[gofs + 64 * r - is];             // zero outside the window
                    const float tj = (float)(64 * r - is) - HALFW + lanef;        // window position - centre
                    de += gw;
                    dq = fmaf(gw, tj * tj, dq);
#endif
                }
            }
            const float e = ok ? ur * ur + ui * ui : 0.0f;
            dpw = fmaf(e, dq, dpw);
            const float s2 = ok ? 2.0f * de : 0.0f;
            vre[r] = s2 * ur;
            vim[r] = -(s2 * ui);
            if (r % 4 == 3) asm volatile("" : "+v"(vre[r]), "+v"(vim[r]), "+v"(dpw));
        } else {
            vre[r] = vim[r] = 0.0f;                                   // circular wrap-around outputs: no gradient
        }
    }
#endif
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                // pooling-row reads done before the next task's DMA
    fft2048w<HALF, kSkip1>(vre, vim, scr, scr_lds, twl, twh, lane);  // g = dL/dS: register i <-> bin 64 brev5(i) + lane
    pin32(vre);
    pin32(vim);
    float amu = 0.0f, asg = 0.0f;
    wg_bwd_tail<DX == 1 ? 1 : 0>(p, A, lane, f, vre, vim, acc_re, acc_im, amu, asg);
    if constexpr (DX == 2) wg_dx_accumulate(p, f, lane, vre, vim, gS, gticket, want);
    amu_out = amu;
    asg_out = asg;
    dpw_out = dpw * (1.0f / (HALFW * HALFW));
}

// the full transposition scratch (fewer LDS store instructions: leaf_fft_wg.hpp) where twelve waves of it fit beside the
// pooling rows -- K = 401 and 201 -- else the half-size column form
constexpr int fft_wg_bwd_row_floats(int SK) { return LEAF_WG_BWD_REGW ? 0 : fft_wg_row_floats(SK); }   // the wave-private pooling row, if any
constexpr size_t fft_wg_bwd_lds_bytes_with(int NW, int SK, int scr_floats) {
    return ((size_t)kTwFloats + 2 * 2 * kWgRingFloat2 + kWgQueueInts + (size_t)NW * (scr_floats + fft_wg_bwd_row_floats(SK))) * 4;
}
constexpr bool fft_wg_bwd_half(int NW, int SK) { return fft_wg_bwd_lds_bytes_with(NW, SK, kWgScrFloats) > (size_t)kMaxLds; }
constexpr size_t fft_wg_bwd_lds_bytes(int NW, int SK) {
    return fft_wg_bwd_lds_bytes_with(NW, SK, fft_wg_bwd_half(NW, SK) ? kWgScrHalfFloats : kWgScrFloats);
}
constexpr int kBlkBwdWaves = 8;                  // leaf_fft_blk_bwd_dx_kernel: two waves per SIMD, 256 VGPRs each
constexpr size_t fft_blk_bwd_lds_bytes(int SK) {
    return ((size_t)kTwFloats + (size_t)kBlkBwdWaves * (2 * kWgRingFloat2 + kWgScrHalfFloats + fft_wg_bwd_row_floats(SK))) * 4;
}

// FftParams fields used beyond the forward's: H = [3][F][2048] real spectra (R | R_mu | R_sigma), gpre, pool_w, dkpart,
// dwpart.
// DX = true: the kernel also yields dL/dx -- G per block in LDS (wg_dx_accumulate / wg_dx_finish above), the block's 2048
// input-gradient samples into part[block][2048]; the transposition scratch is full-size where it fits beside G.
#ifndef LEAF_WG_BWD_DX_FULLSCR
#define LEAF_WG_BWD_DX_FULLSCR 1       // with the pooling rows gone (LEAF_WG_BWD_REGW) the full scratch fits beside G too; 0: half-size (A/B)
#endif
constexpr bool fft_wg_bwd_dx_half(int NW, int SK) {
    return !LEAF_WG_BWD_DX_FULLSCR || fft_wg_bwd_lds_bytes_with(NW, SK, kWgScrFloats) + (size_t)2 * kWgRingFloat2 * 8 > (size_t)kMaxLds;
}
constexpr size_t fft_wg_bwd_dx_lds_bytes(int NW, int SK) {
    return fft_wg_bwd_lds_bytes_with(NW, SK, fft_wg_bwd_dx_half(NW, SK) ? kWgScrHalfFloats : kWgScrFloats) + (size_t)2 * kWgRingFloat2 * 8;
}
template <int SK, int SHOP, int NW, bool DX = false>
__global__ __launch_bounds__(NW * 64, (NW + 3) / 4) void leaf_fft_wg_bwd_kernel(const FftParams p) {
    constexpr bool HALF = DX ? fft_wg_bwd_dx_half(NW, SK) : fft_wg_bwd_half(NW, SK);
    constexpr int SCRF = HALF ? kWgScrHalfFloats : kWgScrFloats;
    extern __shared__ __attribute__((aligned(16))) float wsm[];
    float2* twl = reinterpret_cast<float2*>(wsm);
    float2* twh = twl + 32 * 64;
    float2* ring = twh + 64;                                              // A': [2][kWgRingFloat2]
    [[maybe_unused]] float2* gsum = ring + 2 * kWgRingFloat2;            // DX: G, Hermitian-folded: [2][kWgRingFloat2]
    int* q = reinterpret_cast<int*>(ring + (DX ? 4 : 2) * kWgRingFloat2);
    // q: 0 next task | 1,2 spectra stored per slot | 3,4 inverse tasks finished per slot | 5..8 (clip, block) per slot |
    //    9,10 generations released per slot (all readers done) | 11,12 (DX) filters added to the slot's G, ever
    constexpr int GU = fft_wg_bwd_row_floats(SK);
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane0 = tid & 63;
    float* scr = reinterpret_cast<float*>(q + kWgQueueInts) + (size_t)wave * (SCRF + GU);
    float* sG = scr + SCRF;
    const unsigned scr_lds = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(__attribute__((address_space(3))) float*)scr);

    // band-limited filter tasks (leaf_band_bwd.hpp; parameter gradients only): the plan sits behind the waves' scratch
    constexpr bool BANDK = !HALF && (!DX || LEAF_BAND_BWD_DX) && band_geometry_ok(SK, SHOP) && LEAF_WG_BWD_REGW;   // (DX: the members' shares of the block's G in the task's turn)
    const bool band_on = BANDK && p.band.rec != nullptr;
    int* bl = reinterpret_cast<int*>(wsm + p.band.lds_off);
    if constexpr (BANDK) {
        if (band_on && wave == 0) band_build_plan(p.band.rec, p.band.elist, p.band.n_edge, p.F, bl, lane0, p.band.bias, p.band.smax);
    }
    if (BANDK && band_on) { if (wave > 0) fft_build_twiddles_wg(twl, twh, tid - 64, (NW - 1) * 64); }
    else fft_build_twiddles_wg(twl, twh, tid, NW * 64);
    if (tid < kWgQueueInts) q[tid] = 0;
    __syncthreads();

    constexpr int PADL = SK / 2 + SK % 2 - 1;
    constexpr int LS = fft_block_len(SK, SHOP, true);
    constexpr int DMIN = -((SK - 1 - PADL) / SHOP);
    constexpr int DMAX = (LS - 1 + PADL) / SHOP;
    constexpr int NFR = DMAX - DMIN + 1;
    static_assert(LS % SHOP == 0 && LS % 64 == 0 && LS > 0 && NFR <= 32 && (SK & 1), "static odd-window geometry");

    const int nblocks = p.B * p.nblk;
    const int nset = (nblocks - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x;
    const int NT = band_on ? __builtin_amdgcn_readfirstlane(bl[0]) : p.F;   // filter tasks per block
    const int* tdesc = bl + kBandPlanHead;
    const int* bmem = tdesc + p.F + 4;
    (void)tdesc; (void)bmem;
    const WgTaskGrid grid = wg_task_grid(NT, nset);                        // NT + 1 slots per set
    const int ntasks = nset > 0 ? 1 + nset * (NT + 1) : 0;
    auto pull = [&]() {
        int v = 0;
        if (lane0 == 0) v = __hip_atomic_fetch_add(&q[0], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        return __builtin_amdgcn_readfirstlane(v);
    };
    auto decode = [&](int t, int& set, int& role) { wg_task_decode(grid, t, set, role); };
    // descriptor of a filter task: class (0: one filter on 2048 points, 1 / 2: band task) | index << 2 (filter / first member)
    auto desc_of = [&](int role) {
        if (role <= 0 || role > NT) return 0;
        return band_on ? __builtin_amdgcn_readfirstlane(tdesc[role - 1]) : (role - 1) << 2;
    };
    auto row_of = [&](int role) { return desc_of(role); };
    float rq[32];
    // the 32 table values the task starts with: a spectrum row, or the bins of a band task's windows
    auto load_real_spectrum = [&](int d, int lane) {
        if constexpr (BANDK) {
            if ((d & 3) == 1) { band_load_spectrum<16>(rq, reinterpret_cast<const float*>(p.H), bmem[(d >> 2) + lane / band_lpf(16)], lane); return; }
            if ((d & 3) == 2) { band_load_spectrum<32>(rq, reinterpret_cast<const float*>(p.H), bmem[(d >> 2) + lane / band_lpf(32)], lane); return; }
        }
        const float* src = reinterpret_cast<const float*>(p.H) + (size_t)(d >> 2) * kFftN + lane;
        asm volatile("" ::: "memory");
#pragma unroll
        for (int k = 0; k < 32; ++k) rq[k] = src[64 * k];
        asm volatile("" ::: "memory");
    };

    int seen_set = -1, seen_b = 0, seen_c = 0;                            // block coordinates of the set this wave last worked on
    int t = pull(), set = 0, role = 0;
    if (t < ntasks) decode(t, set, role);
    load_real_spectrum(row_of(role), lane0);
    while (t < ntasks) {
        int lane = lane0;
        asm volatile("" : "+v"(lane));
        const int slot = set & 1, gen = set >> 1;
        float2* A = ring + slot * kWgRingFloat2;
        if (role == 0 || role > NT) {
            if (role == 0 && set < nset) {
                // ---- forward transform of block gb into ring slot `slot`
                const int gb = (int)blockIdx.x + set * (int)gridDim.x;
                const int b = gb / p.nblk, c = gb - b * p.nblk;
                const int n_c = c * LS;
                float are[32], aim[32];
                const float* xb = static_cast<const float*>(p.x) + (size_t)b * p.T;
#pragma unroll
                for (int r = 0; r < 32; ++r) {
                    const int i = 64 * r + lane;
                    const int n = n_c - PADL + ((i + PADL) & (kFftN - 1));
                    are[r] = (n >= 0 && n < p.T) ? xb[n] : 0.0f;
                    aim[r] = 0.0f;
                }
                fft2048w<HALF>(are, aim, scr, scr_lds, twl, twh, lane);
                wg_wait_ge(&q[9 + slot], gen);                            // the slot's previous occupant has been released
#pragma unroll
                for (int i = 0; i < 32; ++i) {
                    const int k = brev5(i);
                    if (k < 16) A[64 * k + lane] = make_float2(are[i], aim[i]);
                    else if (k == 16 && lane == 0) A[1024] = make_float2(are[i], aim[i]);
                }
                if constexpr (DX) {                                       // this block's G starts at zero
                    float2* gS = gsum + slot * kWgRingFloat2;
                    for (int i = lane; i < kWgRingFloat2; i += 64) gS[i] = make_float2(0.0f, 0.0f);
                }
                if (lane == 0) { q[5 + 2 * slot] = b; q[6 + 2 * slot] = c; }
                wg_release();
                if (lane == 0) __hip_atomic_fetch_add(&q[1 + slot], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
            t = pull();
            if (t < ntasks) decode(t, set, role);
            else role = 0;
            load_real_spectrum(row_of(role), lane);
            continue;
        }
        // ---- backward of filter f on the block in ring slot `slot`
        const int tdsc = desc_of(role);
        const int f = tdsc >> 2;
        if (set != seen_set) {                                            // this wave's first filter of the block: once the
            wg_wait_ge(&q[1 + slot], gen + 1);                            // spectrum is in the ring it stays until every filter is done
            seen_b = __builtin_amdgcn_readfirstlane(wg_ld(&q[5 + 2 * slot]));
            seen_c = __builtin_amdgcn_readfirstlane(wg_ld(&q[6 + 2 * slot]));
            seen_set = set;
        }
        const int b = seen_b, c = seen_c;
        const int gb = b * p.nblk + c;
        if constexpr (BANDK) {
            if (tdsc & 3) {
                // ---- band task: the parameter gradients of eight (four) narrow-band filters at the decimated rate (leaf_band_bwd.hpp)
                const int n_c = c * LS;
                const int Lv = min(LS, p.T - n_c);
                int mlo = n_c + PADL - SK + 1;
                mlo = mlo <= 0 ? 0 : (mlo + SHOP - 1) / SHOP;
                const int mhi = min(p.TP - 1, (n_c + Lv - 1 + PADL) / SHOP);
                if ((tdsc & 3) == 1)
                    band_bwd_task<16, SK, SHOP, false, DX>(p, rq, A, bmem + (tdsc >> 2), bl + 4, twl, scr, scr_lds, b, c, gb, mlo, mhi, lane,
                                                           gsum + slot * kWgRingFloat2, &q[11 + slot], gen * NT + role - 1);
                else
                    band_bwd_task<32, SK, SHOP, false, DX>(p, rq, A, bmem + (tdsc >> 2), bl + 4, twl, scr, scr_lds, b, c, gb, mlo, mhi, lane,
                                                           gsum + slot * kWgRingFloat2, &q[11 + slot], gen * NT + role - 1);
                if constexpr (DX) {
                    if (role == NT) wg_dx_finish<HALF>(p, gsum + slot * kWgRingFloat2, nullptr, gb, PADL, lane, scr, scr_lds, twl, twh);
                }
                const int tn_b = pull();
                int nset_b = 0, nrole_b = 0;
                if (tn_b < ntasks) decode(tn_b, nset_b, nrole_b);
                load_real_spectrum(row_of(nrole_b), lane);
                wg_release();
                int old_b = 0;
                if (lane == 0) old_b = __hip_atomic_fetch_add(&q[3 + slot], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                old_b = __builtin_amdgcn_readfirstlane(old_b);
                if (old_b == gen * NT + NT - 1) {
                    if (lane == 0) __hip_atomic_fetch_add(&q[9 + slot], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                }
                t = tn_b;
                set = nset_b;
                role = nrole_b;
                continue;
            }
        }
        float amu, asg, dpw;
        {
            float dummy_re[32], dummy_im[32];
            wg_bwd_filter<SK, SHOP, DX ? 2 : 0, HALF>(p, A, lane, f, b, c, rq, scr, scr_lds, sG, twl, twh, dummy_re, dummy_im, amu, asg,
                                                      dpw, gsum + slot * kWgRingFloat2, &q[11 + slot], gen * NT + role - 1);
        }
        if constexpr (DX) {
            if (role == NT) wg_dx_finish<HALF>(p, gsum + slot * kWgRingFloat2, nullptr, gb, PADL, lane, scr, scr_lds, twl, twh);
        }
        // next task: reserved now, its spectrum row requested before the reductions (rq is free from here)
        const int tn = pull();
        int nset_i = 0, nrole = 0;
        if (tn < ntasks) decode(tn, nset_i, nrole);
        load_real_spectrum(row_of(nrole), lane);
        amu = wave_sum(amu);
        asg = wave_sum(asg);
        dpw = wave_sum(dpw);
        if (lane == 0) {
            const float sp = pool_sigma(p.pool_w[f], SK);
            p.dkpart[((size_t)gb * p.F + f) * 2] = amu;
            p.dkpart[((size_t)gb * p.F + f) * 2 + 1] = asg;
            p.dwpart[(size_t)gb * p.F + f] = dpw / (sp * sp * sp);
        }
        // ---- this task is done with the slot; the wave that finishes the block's last filter releases it
        wg_release();
        int old = 0;
        if (lane == 0) old = __hip_atomic_fetch_add(&q[3 + slot], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        old = __builtin_amdgcn_readfirstlane(old);
        if (old == gen * NT + NT - 1) {
            if (lane == 0) __hip_atomic_fetch_add(&q[9 + slot], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
        t = tn;
        set = nset_i;
        role = nrole;
    }
}

// ---- backward WITH dL/dx: one wave per block (see the header comment).  Blocks are dealt to the persistent waves by
// striding; per block: forward transform -> A' into wave-private LDS; for every filter wg_bwd_filter<.., DX = 1> (which
// adds R_f g into the 64 accumulator registers); then dL/da' = Re(FFT(conj G)), un-rotated into dxblk.
template <int SK, int SHOP>
__global__ __launch_bounds__(kBlkBwdWaves * 64, 2) void leaf_fft_blk_bwd_dx_kernel(const FftParams p) {
    constexpr int SCRF = kWgScrHalfFloats;
    constexpr int GU = fft_wg_bwd_row_floats(SK);
    constexpr int PADL = SK / 2 + SK % 2 - 1;
    constexpr int LS = fft_block_len(SK, SHOP, true);
    extern __shared__ __attribute__((aligned(16))) float wsm[];
    float2* twl = reinterpret_cast<float2*>(wsm);
    float2* twh = twl + 32 * 64;
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane0 = tid & 63;
    float* mine = reinterpret_cast<float*>(twh + 64) + (size_t)wave * (2 * kWgRingFloat2 + SCRF + GU);
    float2* A = reinterpret_cast<float2*>(mine);                          // this wave's block spectrum, bins 0..1024
    float* scr = mine + 2 * kWgRingFloat2;
    float* sG = scr + SCRF;
    const unsigned scr_lds = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(__attribute__((address_space(3))) float*)scr);
    fft_build_twiddles_wg(twl, twh, tid, kBlkBwdWaves * 64);
    __syncthreads();
    // task = (block, group of p.fq filters): the groups exist only to balance the load (B nblk blocks rarely divide evenly
    // over the chip's 2048 wave slots); each yields its own partial input gradient, summed by the gather kernel
    for (int task = blockIdx.x * kBlkBwdWaves + wave; task < p.total_tasks; task += gridDim.x * kBlkBwdWaves) {
        int lane = lane0;
        asm volatile("" : "+v"(lane));
        const int gb = task / p.nfq, fg = task - gb * p.nfq;
        const int f0 = fg * p.fq, f1 = min(p.F, f0 + p.fq);
        const int b = gb / p.nblk, c = gb - b * p.nblk;
        const int n_c = c * LS;
        {
            float are[32], aim[32];
            const float* xb = static_cast<const float*>(p.x) + (size_t)b * p.T;
#pragma unroll
            for (int r = 0; r < 32; ++r) {
                const int i = 64 * r + lane;
                const int n = n_c - PADL + ((i + PADL) & (kFftN - 1));
                are[r] = (n >= 0 && n < p.T) ? xb[n] : 0.0f;
                aim[r] = 0.0f;
            }
            fft2048w<true>(are, aim, scr, scr_lds, twl, twh, lane);
#pragma unroll
            for (int i = 0; i < 32; ++i) {
                const int k = brev5(i);
                if (k < 16) A[64 * k + lane] = make_float2(are[i], aim[i]);
                else if (k == 16 && lane == 0) A[1024] = make_float2(are[i], aim[i]);
            }
        }
        float acc_re[32], acc_im[32];                                     // G = sum_f R_f g_f at bin 64 k + lane
#pragma unroll
        for (int k = 0; k < 32; ++k) acc_re[k] = acc_im[k] = 0.0f;
        for (int f = f0; f < f1; ++f) {
            float rq[32];
            {
                const float* src = reinterpret_cast<const float*>(p.H) + (size_t)f * kFftN + lane;
                asm volatile("" ::: "memory");
#pragma unroll
                for (int k = 0; k < 32; ++k) rq[k] = src[64 * k];
                asm volatile("" ::: "memory");
            }
            float amu, asg, dpw;
            wg_bwd_filter<SK, SHOP, 1>(p, A, lane, f, b, c, rq, scr, scr_lds, sG, twl, twh, acc_re, acc_im, amu, asg, dpw);
            amu = wave_sum(amu);
            asg = wave_sum(asg);
            dpw = wave_sum(dpw);
            if (lane == 0) {
                const float sp = pool_sigma(p.pool_w[f], SK);
                p.dkpart[((size_t)gb * p.F + f) * 2] = amu;
                p.dkpart[((size_t)gb * p.F + f) * 2 + 1] = asg;
                p.dwpart[(size_t)gb * p.F + f] = dpw / (sp * sp * sp);
            }
            pin32(acc_re);
            pin32(acc_im);
        }
        // dL/da'[n] = Re(FFT(conj G))[n]; sample i of the rotated block is x[n_c - padL + ((i + padL) mod N)]
#pragma unroll
        for (int k = 0; k < 32; ++k) acc_im[k] = -acc_im[k];
        fft2048w<true>(acc_re, acc_im, scr, scr_lds, twl, twh, lane);
        float* dst = p.part + (size_t)task * kFftN;
#pragma unroll
        for (int i = 0; i < 32; ++i) dst[(64 * brev5(i) + lane + PADL) & (kFftN - 1)] = acc_re[i];
    }
}

// dL/dx from the per-block input gradients: x[n] belongs to the NB-sample windows (NB = 2048 or 4096) of the blocks c with
// 0 <= n - c L + padL < NB (at most three), each with nfq partials (one per filter group); summed in a fixed order.
#ifndef LEAF_INST_TU               // non-template kernel: compiled once, in leaf_kernels.hip
__global__ void fft_dx_gather_kernel(const float* __restrict__ dxblk, int T, int nblk, int nfq, int L, int padL,
                                     float* __restrict__ dx, int NB = kFftN) {
    // four consecutive samples per thread (a quarter of the waves, four loads in flight each); per sample the same blocks in the
    // same order as one sample per thread would take them
    const int n0 = (blockIdx.x * blockDim.x + threadIdx.x) * 4;
    const int b = blockIdx.y;
    if (n0 >= T) return;
    const int c_hi = min(nblk - 1, (n0 + 3 + padL) / L);
    int c_lo = n0 + padL - (NB - 1);
    c_lo = c_lo <= 0 ? 0 : (c_lo + L - 1) / L;
    float acc[4] = {0.0f, 0.0f, 0.0f, 0.0f};
    for (int c = c_lo; c <= c_hi; ++c) {
        const int i0 = n0 - c * L + padL;                                 // window index of sample n0 in block c
        for (int g = 0; g < nfq; ++g) {
            const float* src = dxblk + (((size_t)b * nblk + c) * nfq + g) * NB;
#pragma unroll
            for (int k = 0; k < 4; ++k)
                if (i0 + k >= 0 && i0 + k < NB) acc[k] += src[i0 + k];
        }
    }
#pragma unroll
    for (int k = 0; k < 4; ++k)
        if (n0 + k < T) dx[(size_t)b * T + n0 + k] = acc[k];
}
#endif

}  // namespace
