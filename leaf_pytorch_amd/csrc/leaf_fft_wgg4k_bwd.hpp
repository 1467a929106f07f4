// leaf_fft_wgg4k_bwd.hpp -- overlap-save BACKWARD on 4096-sample blocks with run-time geometry (odd windows 833..2049):
// the parameter gradients of leaf_fft_wgg_bwd.hpp on the plan of leaf_fft_wgg4k.hpp.
// Part of the single translation unit leaf_kernels.hip (gfx950 only); see that file's header comment.
//
// Per (block, filter), half by half (h = 0: even outputs, h = 1: odd):
//   u_h = the half inverse transform of the forward (zs / zd multiply, 2048 points);
//   pooling backward on the half's samples n_c + 2 k + h with the taps of one parity per frame class
//   (rho = (is_m - h) & 1): d pool_w by gather with taps (j - c)^2 g[j], de_h by scatter -- leaf_fft_wgg_bwd.hpp at half rate;
//   gy_h = 2 de_h y_h, V_h = FFT2048(gy_h).  The 4096-point spectrum of the interleaved gradient is
//   g[e] = V_0[e] + w^e V_1[e], g[e + 2048] = V_0[e] - w^e V_1[e]  (decimation in time), so by linearity each half adds its
//   share of dL/dR[k] = Re(conj(A'[k]) g[k]) to the two spectral dot products as soon as its V_h exists:
//     half 0:  d_lo = Re(conj(A'[e]) V_0[e]),            d_hi =  Re(A'[2048 - e] V_0[e])        (conj(A'[e + 2048]) = A'[2048 - e])
//     half 1:  d_lo = Re(conj(A'[e]) w^e V_1[e]),        d_hi = -Re(A'[2048 - e] w^e V_1[e])
//     d mu += d_lo R_mu[e] + d_hi R_mu[e + 2048],  d sigma likewise  (tables of d w / d mu, d w / d sigma: fft4k_prep_kernel).
// Nothing of size 4096 is ever held: 64 data registers, the wave-private LDS row of the forward, the ring slot.
//
// S801 = true: the static 32 kHz LEAF geometry (K = 801, hop = 320, L = 3200; BASELINE configs[2]).  With an even hop every
// frame of a half meets taps of ONE parity (rho = h), and in half-rate samples the half IS the 16 kHz geometry: sample
// k <-> n = 2 k + h, window start 160 d - 200, taps g_h[i] = g[2 i + h], i = k + 200 - 160 d (401 / 400 taps).  So the pooling
// backward is the register gather of leaf_fft_wg_bwd.hpp (wg_bwd_filter<401, 160>) over 25 rows and 13 frames with the two
// parity rows of the filter (2 x 528 floats, the forward's layout: leaf_fft_wg4k.hpp) DMA'd into wave-private LDS under the
// first transform -- no energy row, no clears, no read-add-write chains; the LDS layout is the static forward's.
#pragma once
#include "leaf_fft_wgg4k.hpp"
#include "leaf_fft_wgg_bwd.hpp"
#include "leaf_band_bwd.hpp"

namespace {

// identity filter -> column map for param_reduce_kernel (the 2048-sample tables get theirs from fft_prep_kernel)
#ifndef LEAF_INST_TU               // non-template kernel: compiled once, in leaf_kernels.hip
__global__ void iota_kernel(int* __restrict__ v, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) v[i] = i;
}
#endif


// ---- dL/dx on 4096-sample blocks (DX = true; the static 32 kHz instance).  dL/dA'[k] = sum_f R_f[k] g_f[k] =: G[k], k < 4096,
// Hermitian-folded as in leaf_fft_wg_bwd.hpp: S[e] = G[e] + conj(G[4096 - e]), e = 0..2048 (S[0] = G[0], S[2048] = conj(G[2048])).
// A filter contributes twice -- once per half, as soon as that half's V_h exists (g[e] = V_0[e] + w^e V_1[e],
// g[e + 2048] = V_0[e] - w^e V_1[e]) -- by plain read-add-write in filter order through a ticket (no float atomics; the sums
// do not depend on timing).  The two halves add to SEPARATE arrays, each with its own chain: on one chain filter i's first half
// (the middle of its task) would wait for filter i - 1's second (the end of that task) and the waves would run two at a time;
// and the evenly staggered waves of a workgroup reach "middle of task j + NW / 2" and "end of task j" at the same moment, which
// one array would serialise.  The first-half array exists per ring slot, so a block's first filters (first halves half a task
// after the previous block's last filter began) never wait for that block's read-out; the second-half array is single (its
// first use comes a whole task later).  Measured alternatives (one array per slot with the halves interleaved at a fixed lag;
// one read-add-write per entry after crossing the mirrored shares between lanes, four ticketed units): 3 .. 7 % slower,
// profiles/r04/ab_4k_dx.txt.
//
// wg4k_dx_accumulate<H>: (vre, vim) = this half's share of g at the low bins, register brev5(k) <-> bin e = 64 k + lane
// (H = 1: already multiplied by w^e); the high bin e + 2048 holds the same value (H = 0) or its negative (H = 1).
// (rlo, rhi)[k] = R_lo[e], R_hi[e], kept in registers since the task's first multiply (two waves per SIMD: the registers are
// there, and a second trip to the tables stands exposed -- 0.9 ms of 7.8 at cfg2).
template <int H>
__device__ __forceinline__ void wg4k_dx_accumulate(const float (&rlo)[32], const float (&rhi)[32], int lane, const float (&vre)[32],
                                                   const float (&vim)[32], float2* S, int* ticket, int want) {
    // bin e = 64 k + lane adds R_lo[e] g to S[e]; bin e + 2048 adds conj(R_hi[e] g') to S[2048 - e].  Chunk C takes rows
    // k = 8 C .. 8 C + 7 of BOTH: the entries its direct rows touch (64 k .. 64 k + 63) and the ones its mirrored rows touch
    // (2048 - 64 k - 63 .. 2048 - 64 k) are disjoint except S[1024] in chunk 2 (row 16, lane 0: its own mirror), so sixteen
    // reads are in flight per step; across chunks the LDS executes a wave's operations in order.
    float2* s1 = S + lane;                                                // S[64 k + lane]
    float2* s2 = S + 64 - lane;                                           // S[2048 - 64 k - lane] = s2[64 (31 - k)]
    auto work = [&](auto cc) {
        constexpr int C = decltype(cc)::value;
        float2 sd[8], sm[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) sd[j] = s1[64 * (8 * C + j)];
#pragma unroll
        for (int j = 0; j < 8; ++j) sm[j] = s2[64 * (31 - (8 * C + j))];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int k = 8 * C + j;
            const float a = rlo[k];
            sd[j].x = fmaf(a, vre[brev5(k)], sd[j].x);
            sd[j].y = fmaf(a, vim[brev5(k)], sd[j].y);
            if (k == 16 && lane == 0) sm[j] = sd[j];                      // S[1024]: the mirrored share adds to the direct one
            const float r = H ? -rhi[k] : rhi[k];
            sm[j].x = fmaf(r, vre[brev5(k)], sm[j].x);
            sm[j].y = fmaf(-r, vim[brev5(k)], sm[j].y);
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) s1[64 * (8 * C + j)] = sd[j];
#pragma unroll
        for (int j = 0; j < 8; ++j) s2[64 * (31 - (8 * C + j))] = sm[j];     // (after the direct stores: S[1024] keeps both shares)
    };
#ifndef LEAF_DX_NOWAIT                 // measurement only (wrong sums): what the ordered turn costs
    wg_wait_ge(ticket, want);                                             // the previous filter's share of this half is in
#endif
    if (LEAF_DX_PRIO) __builtin_amdgcn_s_setprio(3);                      // (leaf_fft_wg_bwd.hpp: the turn's holder goes first)
    work(std::integral_constant<int, 0>{});
    work(std::integral_constant<int, 1>{});
    work(std::integral_constant<int, 2>{});
    work(std::integral_constant<int, 3>{});
    wg_release();
    if (lane == 0) __hip_atomic_fetch_add(ticket, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    if (LEAF_DX_PRIO) __builtin_amdgcn_s_setprio(0);
}
// wg4k_dx_finish: called by the wave that added the block's last share (the second half of the last filter in queue order;
// both chains add in that order, so every other share is in).  S = S0 + S1;  X = the Hermitian spectrum whose 4096-point
// transform is dL/da': X[e] = conj(S[e]) / 2, X[e + 2048] = S[2048 - e] / 2 (0 < e < 2048), X[0] = Re S[0], X[2048] = Re S[2048].
// One 2048-point transform yields the real 4096 samples: with E[e] = X[e] + X[e + 2048] (even samples) and
// O[e] = (X[e] - X[e + 2048]) w^e (odd samples), FFT2048(E + i O)[m] = dx[2 m] + i dx[2 m + 1].  Sample i of the rotated
// block is x[n_c - padL + ((i + rot) mod 4096)]: stored un-rotated into part[gb][4096] (rot is even: pairs stay together).
// Clears both arrays for their next blocks.
__device__ __forceinline__ void wg4k_dx_finish(const FftParams& p, float2* S0, float2* S1, int gb, int rot, int lane, float* scr,
                                               unsigned scr_lds, const float2* twl, const float2* twh, const float2* tw4a,
                                               const float2* tw4b) {
    float zre[32], zim[32];
    const unsigned d0 = lds_addr(S0 + lane), m0 = lds_addr(S0 + 64 - lane);
    const unsigned d1 = lds_addr(S1 + lane), m1 = lds_addr(S1 + 64 - lane);
    const float2 wl = tw4b[lane];
    auto chunk = [&](auto cc) {
        constexpr int C = decltype(cc)::value;
        v2f a0[8], b0[8];
        {
            v2f a1[8], b1[8];
            wg4k_ring_chunk<C>(a0, b0, d0, m0);                           // S[e], S[2048 - e]
            wg4k_ring_chunk<C>(a1, b1, d1, m1);
#pragma unroll
            for (int j = 0; j < 8; ++j) { a0[j].x += a1[j].x; a0[j].y += a1[j].y; b0[j].x += b1[j].x; b0[j].y += b1[j].y; }
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int k = 8 * C + j;
            const bool self = k == 0 && lane == 0;
            const float xr = self ? a0[j].x : 0.5f * a0[j].x, xi = self ? 0.0f : -0.5f * a0[j].y;   // X[e]
            const float yr = self ? b0[j].x : 0.5f * b0[j].x, yi = self ? 0.0f : 0.5f * b0[j].y;    // X[e + 2048]
            const float er = xr + yr, ei = xi + yi, dr = xr - yr, di = xi - yi;
            const float2 wk = tw4a[k];
            const float wr = wk.x * wl.x - wk.y * wl.y, wi = wk.x * wl.y + wk.y * wl.x;  // w^(64 k + lane)
            const float orr = dr * wr - di * wi, oi = dr * wi + di * wr;
            zre[k] = er - oi;
            zim[k] = ei + orr;
        }
        asm volatile("" : "+v"(zre[8 * C]), "+v"(zre[8 * C + 1]), "+v"(zre[8 * C + 2]), "+v"(zre[8 * C + 3]),
                          "+v"(zre[8 * C + 4]), "+v"(zre[8 * C + 5]), "+v"(zre[8 * C + 6]), "+v"(zre[8 * C + 7]),
                          "+v"(zim[8 * C]), "+v"(zim[8 * C + 1]), "+v"(zim[8 * C + 2]), "+v"(zim[8 * C + 3]),
                          "+v"(zim[8 * C + 4]), "+v"(zim[8 * C + 5]), "+v"(zim[8 * C + 6]), "+v"(zim[8 * C + 7]));
    };
    chunk(std::integral_constant<int, 0>{}); chunk(std::integral_constant<int, 1>{});
    chunk(std::integral_constant<int, 2>{}); chunk(std::integral_constant<int, 3>{});
    {
        const float2 zero = make_float2(0.0f, 0.0f);
        float2* z0 = S0 + lane;
        float2* z1 = S1 + lane;
#pragma unroll
        for (int k = 0; k < 32; ++k) { z0[64 * k] = zero; z1[64 * k] = zero; }
        if (lane == 0) { S0[2048] = zero; S1[2048] = zero; }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    fft2048w<true>(zre, zim, scr, scr_lds, twl, twh, lane);
    float2* dst = reinterpret_cast<float2*>(p.part + (size_t)gb * kFft4N);
#pragma unroll
    for (int i = 0; i < 32; ++i) {
        const int m = 64 * brev5(i) + lane;
        dst[((2 * m + rot) & (kFft4N - 1)) >> 1] = make_float2(zre[i], zim[i]);
    }
}

template <int NW, int NI2, bool S801 = false, bool DX = false>
__global__ __launch_bounds__(NW * 64, (NW + 3) / 4) void leaf_fft_wgg4k_bwd_kernel(const FftParams p) {
    using gfp = const __attribute__((address_space(1))) float*;          // table pointers that stay `global` when made opaque
    using gf2p = const __attribute__((address_space(1))) v2f*;            // (a builtin vector: HIP's float2 class does not load through address spaces)
    static_assert(!DX || S801, "dL/dx on 4096-sample blocks: the static 32 kHz instance only");
    constexpr bool FULLSCR = S801 && !DX && LEAF_4K_BWD_REGW && LEAF_4K_BWD_FULLSCR;    // full transposition scratch, no rows
    constexpr bool HS = !FULLSCR;
    constexpr int S801_WAVE_FLOATS = FULLSCR ? kWgScrFloats : kWgScrHalfFloats + fft_wg4k_bwd_rows(DX) * kWg4RowFloats;
    extern __shared__ __attribute__((aligned(16))) float wsm[];
    float2* twl = reinterpret_cast<float2*>(wsm);                        // [32][64]
    float2* twh = twl + 32 * 64;                                          // [32][2]
    float2* tw4a = twh + 64;                                              // w^(64 k), k < 32
    float2* tw4b = tw4a + 32;                                             // w^lane
    float2* ring = tw4b + 64;                                             // [2][kWg4RingFloat2]
    int* q = reinterpret_cast<int*>(ring + 2 * kWg4RingFloat2);
    const int PF = S801 ? 0 : fft_wgg4k_front_floats(p.K), BP = S801 ? 0 : fft_wgg4k_back_floats(p.K);
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane0 = tid & 63;
    // S801: [transposition scratch | the filter's two parity pooling rows] per wave (fft_wg4k_bwd_lds_bytes)
    float* wbase = reinterpret_cast<float*>(q + kWgQueueInts) +
                   (size_t)wave * (S801 ? S801_WAVE_FLOATS : PF + kFftN + BP + p.NT);   // p.NT: frame-sum floats
    float* scr = wbase + PF;                                              // energies [0, 2048); transposition scratch in its head
    [[maybe_unused]] float* sG = scr + kWgScrHalfFloats;                 // (S801)
    // (DX) behind the per-wave areas (transposition scratch; with LEAF_4K_BWD_REGW = 0 also one pooling row, fetched per half) the folded gradient
    // spectra -- first-half shares per ring slot [0], [1], second-half shares [2] -- and their tickets (fft_wg4k_bwd_dx_lds_bytes)
    [[maybe_unused]] float2* gsum = reinterpret_cast<float2*>(reinterpret_cast<float*>(q + kWgQueueInts) +
                                                              (size_t)NW * S801_WAVE_FLOATS);
    [[maybe_unused]] int* gtick = reinterpret_cast<int*>(gsum + 3 * kWg4RingFloat2);   // shares added + read-outs, ever, per array
    const unsigned scr_lds = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(__attribute__((address_space(3))) float*)scr);

    // band-limited filter tasks of the static 32 kHz instance (leaf_band_bwd.hpp, N4K; parameter gradients only)
    static_assert(kFft4TabFloats == 12288, "leaf_band_bwd.hpp addresses the derivative tables of the 4096-sample plan by this stride");
    constexpr bool BANDK = S801 && !DX && FULLSCR;
    const bool band_on = BANDK && p.band.rec != nullptr;
    int* bl = reinterpret_cast<int*>(wsm + p.band.lds_off);
    if constexpr (BANDK) {
        if (band_on && wave == 0) band_build_plan(p.band.rec, p.band.elist, p.band.n_edge, p.F, bl, lane0, p.band.bias, p.band.smax);
    }
    fft_build_twiddles_wg(twl, twh, tid, (int)blockDim.x);
    for (int i = tid; i < 96; i += (int)blockDim.x) {
        float s, c;
        sincospif(2.0f * (float)(i < 32 ? 64 * i : i - 32) / (float)kFft4N, &s, &c);
        tw4a[i] = make_float2(c, -s);                                     // (tw4b follows tw4a contiguously)
    }
    if (tid < kWgQueueInts) q[tid] = 0;
    if constexpr (DX) {
        for (int i = tid; i < 3 * kWg4RingFloat2; i += (int)blockDim.x) gsum[i] = make_float2(0.0f, 0.0f);
        if (tid < 8) gtick[tid] = 0;
    }
    if constexpr (!S801) {
        for (int i = lane0; i < PF; i += 64) wbase[i] = 0.0f;             // written once: nothing else touches the paddings
        for (int i = lane0; i < BP; i += 64) scr[kFftN + i] = 0.0f;
    }
    __syncthreads();

    // odd windows: ROT = PADL
    const int PADL = S801 ? 400 : p.padL, ROT = S801 ? 400 : p.K / 2, LS = S801 ? 3200 : p.L, SKr = S801 ? 801 : p.K,
              SHOPr = S801 ? 320 : p.hop;

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
                const int gb = (int)blockIdx.x + set * (int)gridDim.x;
                const int b = gb / p.nblk, c = gb - b * p.nblk;
                const int n_c = c * LS;
                const float* xb = static_cast<const float*>(p.x) + (size_t)b * p.T;
                const unsigned short* xh = static_cast<const unsigned short*>(p.x) + (size_t)b * p.T;
                auto sample = [&](int i) -> float {                       // rotated block a'[i] = xz[n_c - padL + ((i + padL) mod 4096)]
                    const int n = n_c - PADL + ((i + ROT) & (kFft4N - 1));
                    if (p.io_bf16) {
                        const unsigned v = xh[min(max(n, 0), p.T - 1)];
                        return (n >= 0 && n < p.T) ? __uint_as_float(v << 16) : 0.0f;
                    }
                    return (n >= 0 && n < p.T) ? xb[n] : 0.0f;
                };
                float xre[32], xim[32];
#pragma unroll
                for (int r = 0; r < 32; ++r) { xre[r] = sample(2 * (64 * r + lane)); xim[r] = 0.0f; }
                fft2048w<HS>(xre, xim, scr, scr_lds, twl, twh, lane);
                wg_wait_ge(&q[9 + slot], gen);                            // the slot's previous occupant has been released
#pragma unroll
                for (int i = 0; i < 32; ++i) A[64 * brev5(i) + lane] = make_float2(xre[i], xim[i]);
#pragma unroll
                for (int r = 0; r < 32; ++r) { xre[r] = sample(2 * (64 * r + lane) + 1); xim[r] = 0.0f; }
                fft2048w<HS>(xre, xim, scr, scr_lds, twl, twh, lane);
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
        // ---- backward of filter f on the block in ring slot `slot` (odd sets walk the tasks backwards: leaf_fft_wg4k.hpp)
        const int ti_ = (LEAF_SWEEP_BACK && (set & 1)) ? NT - role : role - 1;
        const int tdsc = band_on ? __builtin_amdgcn_readfirstlane(tdesc[ti_]) : ti_ << 2;   // class (0: one filter; 2: band task) | index << 2
        const int f = tdsc >> 2;
        // this filter's tables, wave-uniform bases (tab_ld: base + the lane's byte offset + an immediate):
        // (R_lo, R_hi)[2048] f2 | (D_lo, D_hi)[2048] f4 of w; the mu slab's second part: (d/dmu lo, hi, d/dsigma lo, hi)[2048] f4
        const float* Rtab = reinterpret_cast<const float*>(p.H) + (size_t)f * kFft4TabFloats;
        const float* Mtab = reinterpret_cast<const float*>(p.H) + ((size_t)p.F + f) * kFft4TabFloats;
        const float* Stab = reinterpret_cast<const float*>(p.H) + ((size_t)2 * p.F + f) * kFft4TabFloats;
        const unsigned lane4 = 4u * (unsigned)lane;
        if (set != seen_set) {                                            // this wave's first filter of the block: once the
            wg_wait_ge(&q[1 + slot], gen + 1);                            // spectrum is in the ring it stays until every filter is done
            seen_b = __builtin_amdgcn_readfirstlane(wg_ld(&q[5 + 2 * slot]));
            seen_c = __builtin_amdgcn_readfirstlane(wg_ld(&q[6 + 2 * slot]));
            seen_set = set;
        }
        const int b = seen_b, c = seen_c;
        const int gb = b * p.nblk + c;
        const int n_c = c * LS;
        const int Lv = min(LS, p.T - n_c);
        int mlo = n_c + PADL - SKr + 1;
        mlo = mlo <= 0 ? 0 : (mlo + SHOPr - 1) / SHOPr;
        const int mhi = min(p.TP - 1, (n_c + Lv - 1 + PADL) / SHOPr);
        if constexpr (BANDK) {
            if (tdsc & 3) {
                // ---- band task: the parameter gradients of four narrow-band filters at the decimated rate (leaf_band_bwd.hpp); the
                // filter's values for bins kb + j are R[4096 - kb - j] = R_hi[2048 - kb - j] (leaf_fft_wg4k_kernel)
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
                band_bwd_task<32, 801, 320, true>(p, rq, A, mem, bl + 4, twl, scr, scr_lds, b, c, gb, mlo, mhi, lane);
                const int tn_b = pull();
                int nset_b = 0, nrole_b = 0;
                if (tn_b < ntasks) decode(tn_b, nset_b, nrole_b);
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
        // S801: half-rate geometry of both halves (see the header comment) and g_pre of the block's frames as scalars
        constexpr int kHHop = 160, kHPad = 200, kHK = 401, kHRows = 3200 / 2 / 64;
        constexpr int kDMin = -((kHK - 1 - kHPad) / kHHop), kDMax = (3200 / 2 - 1 + kHPad) / kHHop, kNFr = kDMax - kDMin + 1;
        [[maybe_unused]] float gp[kNFr];
        if constexpr (S801) {
            // the filter's two parity rows -> wave-private LDS (the previous task's reads of them are complete: it ended
            // with s_waitcnt lgkmcnt(0)); they land under the first transform
            // (DX: one row buffer -- the first half's row now, the second's once the first half has read it: fetch_row below)
#if !LEAF_4K_BWD_REGW
            const float* gsrc = p.Gz + (size_t)f * 2 * kWg4RowFloats;
            constexpr int GU2 = (DX ? 1 : 2) * kWg4RowFloats;
#pragma unroll
            for (int i0 = 0; i0 < GU2; i0 += 256)
                if (i0 + 256 <= GU2 || i0 + 4 * lane < GU2)
                    __builtin_amdgcn_global_load_lds(gsrc + i0 + 4 * lane, (__attribute__((address_space(3))) void*)(sG + i0), 16, 0, 0);
#endif
            asm volatile("" ::: "memory");
            const int fi = lane & 31, m = n_c / SHOPr + kDMin + fi;
            const float mine = (fi < kNFr && m >= mlo && m <= mhi) ? p.gpre[((size_t)b * p.F + f) * p.TP + m] : 0.0f;
#pragma unroll
            for (int qq = 0; qq < kNFr; ++qq) gp[qq] = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(mine), qq));
        }
        const unsigned a_dir = lds_addr(A + lane), a_mir = lds_addr(A + (2048 - 64 * 31) - lane);
        // g_pre of the frames this block meets (at most 64: the host checks), one per lane; read back with v_readlane
        const float* gp_row = p.gpre + ((size_t)b * p.F + f) * p.TP;
        const float gp_mine = !S801 && mlo + lane <= mhi ? gp_row[mlo + lane] : 0.0f;
        const float half = 0.5f * (float)(SKr - 1);
        float qacc = 0.0f, amu = 0.0f, asg = 0.0f;
        float zre[32], zim[32];
        [[maybe_unused]] float rlo_k[DX ? 32 : 1], rhi_k[DX ? 32 : 1];    // (DX) R_lo / R_hi at this lane's bins, for the task
        using lds_fp = __attribute__((address_space(3))) float*;
        using lds_cfp = const __attribute__((address_space(3))) float*;
        using f4 = float __attribute__((ext_vector_type(4)));
        using lds_f4p = __attribute__((address_space(3))) f4*;
        const f4 zero4 = {0.0f, 0.0f, 0.0f, 0.0f};
        const lds_fp erow = (lds_fp)scr + lane;
        const lds_f4p zrow = (lds_f4p)wbase + lane;
        // pooling backward + second transform + spectral share of one half; (zre, zim) hold u_h on entry
#if LEAF_4K_BWD_REGW
        // S801: the half's parity row as NJ register vectors, pw[k][lane] = row[PJ0 + PG k + lane]
        constexpr int PG = wg_pool_step(kHHop), PJ0 = wg_pool_jmin(kHK, kHHop), NJ = wg_pool_nj(kHK, kHHop);
        [[maybe_unused]] float pw[S801 ? NJ : 1];
        [[maybe_unused]] auto load_pw = [&](int h) {
            if constexpr (S801) {
                const float* wsrc = p.Gz + ((size_t)f * 2 + h) * kWg4RowFloats + (kGPad + PJ0) + lane;
                asm volatile("" ::: "memory");
#pragma unroll
                for (int k = 0; k < NJ; ++k) pw[k] = wsrc[PG * k];
                asm volatile("" ::: "memory");
            }
        };
#endif
        auto half_bwd = [&](auto hh) {
            constexpr int h = decltype(hh)::value;
            pin32(zre);
            pin32(zim);
            float vre[32], vim[32];
            if constexpr (S801) {
                // pooling backward by register gather, row by row (64 half-rate samples each): de = sum over the frames whose
                // window meets the row of g_pre[m] g_h[i], dq the same with (j - centre)^2, j = 2 i + h
#if LEAF_4K_BWD_REGW
                load_pw(h);                                               // (requesting them before the transform measured the same)
#if LEAF_4K_BWD_PW2
                float pw2[NJ];                                            // the weights times (full-rate tap index - centre)^2: d pool_w
#pragma unroll
                for (int k = 0; k < NJ; ++k) {
                    const float tj = (float)(2 * (PJ0 + PG * k) + h - 400) + 2.0f * (float)lane;
                    pw2[k] = pw[k] * (tj * tj);
                }
#endif
#else
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // the rows' DMA has landed (g_pre loads with it)
                const float* sGh = sG + (DX ? 0 : h) * kWg4RowFloats;
#endif
#if LEAF_4K_BWD_REGW && LEAF_4K_BWD_PW2 && LEAF_4K_BWD_FUSE2
                // rows r and r + 16 together, and with them the first decimation-in-time stage of the transform that follows
                // (leaf_fft_wg_bwd.hpp, LEAF_WG_BWD_FUSE2): out[r] = gy[r] + gy[r + 16], out[r + 16] = gy[r] - gy[r + 16]
                auto row_grad = [&](auto rr, float& s2, float& ur, float& ui) {
                    constexpr int r = decltype(rr)::value;
                    ur = zre[brev5(r)];
                    ui = zim[brev5(r)];
                    const bool ok = 2 * (64 * r + lane) + h < Lv;
                    float de = 0.0f, dq = 0.0f;
#pragma unroll
                    for (int fi = 0; fi < kNFr; ++fi) {
                        const int is = (kDMin + fi) * kHHop - kHPad;
                        if (is <= 64 * r + 63 && is + kHK > 64 * r) {
                            de = fmaf(gp[fi], pw[(64 * r - is - PJ0) / PG], de);
                            dq = fmaf(gp[fi], pw2[(64 * r - is - PJ0) / PG], dq);
                        }
                    }
                    const float e = ok ? ur * ur + ui * ui : 0.0f;
                    qacc = fmaf(e, dq, qacc);
                    s2 = ok ? 2.0f * de : 0.0f;
                };
                auto row_pair = [&](auto rr) {
                    constexpr int r = decltype(rr)::value;
                    float are = 0.0f, aim = 0.0f;
                    if constexpr (r < kHRows) {
                        float s2a, ura, uia;
                        row_grad(rr, s2a, ura, uia);
                        are = s2a * ura;
                        aim = -(s2a * uia);
                    }
                    if constexpr (r + 16 < kHRows) {
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
                row_pair(std::integral_constant<int, B0 + 0>{}); row_pair(std::integral_constant<int, B0 + 1>{});        \
                row_pair(std::integral_constant<int, B0 + 2>{}); row_pair(std::integral_constant<int, B0 + 3>{});        \
                asm volatile("" : "+v"(vre[B0 + 3]), "+v"(vim[B0 + 3]), "+v"(vre[B0 + 19]), "+v"(vim[B0 + 19]), "+v"(qacc));
                LEAF_ROW4(0) LEAF_ROW4(4) LEAF_ROW4(8) LEAF_ROW4(12)
#undef LEAF_ROW4
#else
                const float lane2 = 2.0f * (float)lane;
                int gofs = kGPad + lane;                                  // made opaque per row group: keeps the rows in program order
#pragma unroll
                for (int r = 0; r < 32; ++r) {
                    const int i = brev5(r);                               // register holding row r of u_h
                    if (r < kHRows) {
                        const float ur = zre[i], ui = zim[i];
                        const bool ok = 2 * (64 * r + lane) + h < Lv;
                        float de = 0.0f, dq = 0.0f;
                        if (r % 4 == 0) asm volatile("" : "+v"(gofs));
#pragma unroll
                        for (int fi = 0; fi < kNFr; ++fi) {
                            const int is = (kDMin + fi) * kHHop - kHPad;  // half-rate window start relative to the block
                            if (is <= 64 * r + 63 && is + kHK > 64 * r) {
#if LEAF_4K_BWD_REGW && LEAF_4K_BWD_PW2
                                de = fmaf(gp[fi], pw[(64 * r - is - PJ0) / PG], de);       // zero outside the window
                                dq = fmaf(gp[fi], pw2[(64 * r - is - PJ0) / PG], dq);      // the same weight times (tap - centre)^2
#else
#if LEAF_4K_BWD_REGW
                                const float gw = gp[fi] * pw[(64 * r - is - PJ0) / PG];    // zero outside the window
#else
                                const float gw = gp[fi] * sGh[gofs + 64 * r - is];           // zero outside the window
#endif
                                const float tj = (float)(2 * (64 * r - is) + h - 400) + lane2;   // full-rate tap index - centre
                                de += gw;
                                dq = fmaf(gw, tj * tj, dq);
#endif
                            }
                        }
                        const float e = ok ? ur * ur + ui * ui : 0.0f;
                        qacc = fmaf(e, dq, qacc);
                        const float s2 = ok ? 2.0f * de : 0.0f;
                        vre[r] = s2 * ur;
                        vim[r] = -(s2 * ui);
                        if (r % 4 == 3) asm volatile("" : "+v"(vre[r]), "+v"(vim[r]), "+v"(qacc));
                    } else {
                        vre[r] = vim[r] = 0.0f;                           // beyond the block's 3200 samples: no gradient
                    }
                }
#endif
            } else {
            // |y|^2 -> the row; the paddings are cleared too (the previous scatter ran into them)
            for (int i0 = 0; i0 < PF; i0 += 256)
                if (i0 + 4 * lane < PF) zrow[i0 / 4] = zero4;
            for (int i0 = 0; i0 < BP; i0 += 256)
                if (i0 + 4 * lane < BP) zrow[(PF + kFftN + i0) / 4] = zero4;
#pragma unroll
            for (int i = 0; i < 32; ++i) {
                const int r = brev5(i);
                erow[64 * r] = 2 * (64 * r + lane) + h < Lv ? zre[i] * zre[i] + zim[i] * zim[i] : 0.0f;
            }
            const int step = (SHOPr & 1) ? 2 : 1;
            const int rho_lo = (mlo * SHOPr - PADL - n_c - h) & 1;        // tap parity of frame mlo in this half
            // taps of parity rho: lane l holds g[2 (64 i + l) + rho]
            auto load_taps = [&](int rho, float (&w)[NI2]) {
                gfp q = (gfp)(p.Gz + (size_t)f * 2 * p.GZ + kGPad + (rho ? p.GZ : 0)) + lane;
                asm volatile("" : "+v"(q) : : "memory");
#pragma unroll
                for (int i = 0; i < NI2; ++i) w[i] = q[64 * i];
            };
            // frames of one parity class: m_first, m_first + step, ...
            auto first_of = [&](int rho) { return step == 1 ? (rho == rho_lo ? mlo : mhi + 1) : mlo + (rho == rho_lo ? 0 : 1); };
            // (a) d pool_w: gather with the taps times (j - centre)^2, j = 2 (64 i + l) + rho
            for (int rho = 0; rho < 2; ++rho) {
                const int m_first = first_of(rho);
                if (m_first > mhi) continue;
                float w2[NI2];
                load_taps(rho, w2);
                {
                    float tj = (float)(2 * lane + rho) - half;
#pragma unroll
                    for (int i = 0; i < NI2; ++i) {
                        w2[i] *= tj * tj;
                        tj += 128.0f;
                    }
                }
                const lds_cfp ebase = (lds_cfp)scr + lane;
                const int m_last = m_first + (mhi - m_first) / step * step;
#pragma nounroll
                for (int m4 = m_first; m4 <= mhi; m4 += 4 * step) {       // four frames per turn: independent FMA chains
                    float a[4];
                    lds_cfp pk[4];
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const int m = min(m4 + k * step, m_last);
                        pk[k] = ebase + ((m * SHOPr - PADL - n_c - h + 1) >> 1);
                        a[k] = 0.0f;
                    }
#pragma unroll
                    for (int i = 0; i < NI2; ++i)
#pragma unroll
                        for (int k = 0; k < 4; ++k) a[k] = fmaf(w2[i], pk[k][64 * i], a[k]);
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const int m = m4 + k * step;
                        const float gpm = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(gp_mine), min(m, mhi) - mlo));
                        qacc = fmaf(m <= mhi ? gpm : 0.0f, a[k], qacc);
                    }
                }
            }
            // (b) de_h[k] = sum_m g_pre[m] g[2 k + h - is_m]: clear the row, scatter frame by frame (read, add, write; the
            //     LDS executes a wave's operations in order)
            for (int i0 = 0; i0 < kFftN; i0 += 256) zrow[(PF + i0) / 4] = zero4;
            for (int rho = 0; rho < 2; ++rho) {
                const int m_first = first_of(rho);
                if (m_first > mhi) continue;
                float w[NI2];
                load_taps(rho, w);
                const lds_fp dbase = (lds_fp)scr + lane;
#pragma nounroll
                for (int m = m_first; m <= mhi; m += step) {
                    const float gpm = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(gp_mine), m - mlo));
                    const int isx = m * SHOPr - PADL - n_c - h;
                    const lds_fp pk = dbase + ((isx + 1) >> 1);
                    float tv[NI2];
#pragma unroll
                    for (int i = 0; i < NI2; ++i) tv[i] = pk[64 * i];
#pragma unroll
                    for (int i = 0; i < NI2; ++i) pk[64 * i] = fmaf(gpm, w[i], tv[i]);
                }
            }
            // (c) gy_h = 2 de_h y_h (natural row order), second transform
#pragma unroll
            for (int r = 0; r < 32; ++r) {
                const int i = brev5(r);
                const float de = erow[64 * r];
                const float s2 = 2 * (64 * r + lane) + h < Lv ? 2.0f * de : 0.0f;
                vre[r] = s2 * zre[i];
                vim[r] = -(s2 * zim[i]);
            }
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");            // the row's reads are done before the transform's scratch writes
            if constexpr (S801 && DX && h == 0 && !LEAF_4K_BWD_REGW) {
                // the second half's parity row into the same buffer: it lands under the transform below
                const float* gsrc1 = p.Gz + ((size_t)f * 2 + 1) * kWg4RowFloats;
#pragma unroll
                for (int i0 = 0; i0 < kWg4RowFloats; i0 += 256)
                    if (i0 + 256 <= kWg4RowFloats || i0 + 4 * lane < kWg4RowFloats)
                        __builtin_amdgcn_global_load_lds(gsrc1 + i0 + 4 * lane, (__attribute__((address_space(3))) void*)(sG + i0), 16, 0, 0);
                asm volatile("" ::: "memory");
            }
            fft2048w<HS, S801 && LEAF_4K_BWD_REGW && LEAF_4K_BWD_PW2 && LEAF_4K_BWD_FUSE2>(vre, vim, scr, scr_lds, twl, twh, lane);   // V_h: register brev5(k) <-> bin 64 k + lane
            pin32(vre);
            pin32(vim);
            // (d) this half's share of the spectral dot products
            const float2 wl = tw4b[lane];
            auto chunk = [&](auto cc) {
                constexpr int C = decltype(cc)::value;                    // bins 64 k + lane, k = 8 C .. 8 C + 7
                float ml[8], mh[8], sl[8], sh8[8];
                // the lane's byte offset is made opaque HERE (the loads cannot issue earlier); base + offset + immediate costs no
                // address arithmetic (an opaque element INDEX cost three VALU instructions per load -- ~1 000 of the task's ~10 000)
                unsigned vo = lane4;
                asm volatile("" : "+v"(vo), "+v"(amu), "+v"(asg) : : "memory");     // the previous chunk is complete
                using f4 = float __attribute__((ext_vector_type(4)));
                const unsigned vo4 = 4u * vo;
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const f4 t = tab_ld<f4>(Mtab, vo4, 16384 + 1024 * (8 * C + j));   // (mu lo, mu hi, sigma lo, sigma hi) of bin 64 k + lane
                    ml[j] = t.x; mh[j] = t.y; sl[j] = t.z; sh8[j] = t.w;
                }
                (void)Stab;
                asm volatile("" ::: "memory");
                v2f a[8], m[8];
                wg4k_ring_chunk<C>(a, m, a_dir, a_mir);
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const int k = 8 * C + j;
                    float gr = vre[brev5(k)], gi = vim[brev5(k)];
                    if (h == 1) {                                         // w^e V_1[e], w^e = (c, -s) = tw4a[k] * tw4b[lane]
                        const float2 wk = tw4a[k];
                        const float wr = wk.x * wl.x - wk.y * wl.y, wi = wk.x * wl.y + wk.y * wl.x;
                        const float tr = gr * wr - gi * wi, ti = gr * wi + gi * wr;
                        gr = tr;
                        gi = ti;
                        if constexpr (DX) { vre[brev5(k)] = gr; vim[brev5(k)] = gi; }   // kept for the block's G below
                    }
                    const float d_lo = a[j].x * gr + a[j].y * gi;                    // Re(conj(A'[e]) g)
                    float d_hi = m[j].x * gr - m[j].y * gi;                          // Re(A'[2048 - e] g)
                    if (h == 1) d_hi = -d_hi;
                    amu = fmaf(d_lo, ml[j], fmaf(d_hi, mh[j], amu));
                    asg = fmaf(d_lo, sl[j], fmaf(d_hi, sh8[j], asg));
                }
            };
            chunk(std::integral_constant<int, 0>{}); chunk(std::integral_constant<int, 1>{});
            chunk(std::integral_constant<int, 2>{}); chunk(std::integral_constant<int, 3>{});
            asm volatile("" : "+v"(amu), "+v"(asg), "+v"(qacc) : : "memory");
            // (e) DX: this half's share of the block's G, in the order the queue hands the filters out
#ifndef LEAF_4K_DX_NOACC               // measurement only (wrong dL/dx): what the accumulation costs
            if constexpr (DX)
                wg4k_dx_accumulate<h>(rlo_k, rhi_k, lane, vre, vim, gsum + (h ? 2 : slot) * kWg4RingFloat2, gtick + (h ? 2 : slot),
                                      (h ? set : gen) * (p.F + 1) + role - 1);
#endif
        };
        // ---- even output samples: zs = conj(A'[e]) R_lo[e] + A'[2048 - e] R_hi[e]
        {
            auto chunk = [&](auto cc) {
                constexpr int C = decltype(cc)::value;
                float rl[8], rh[8];
                // the table offset is made opaque HERE: the loads below cannot issue before this point (a plain "memory"
                // clobber does not hold them -- they are hoisted under the previous phase and spilled one by one)
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
                    if constexpr (DX) { rlo_k[k] = rl[j]; rhi_k[k] = rh[j]; }
                }
                asm volatile("" : "+v"(zre[8 * C]), "+v"(zre[8 * C + 1]), "+v"(zre[8 * C + 2]), "+v"(zre[8 * C + 3]),
                                  "+v"(zre[8 * C + 4]), "+v"(zre[8 * C + 5]), "+v"(zre[8 * C + 6]), "+v"(zre[8 * C + 7]),
                                  "+v"(zim[8 * C]), "+v"(zim[8 * C + 1]), "+v"(zim[8 * C + 2]), "+v"(zim[8 * C + 3]),
                                  "+v"(zim[8 * C + 4]), "+v"(zim[8 * C + 5]), "+v"(zim[8 * C + 6]), "+v"(zim[8 * C + 7]));
            };
            chunk(std::integral_constant<int, 0>{}); chunk(std::integral_constant<int, 1>{});
            chunk(std::integral_constant<int, 2>{}); chunk(std::integral_constant<int, 3>{});
        }
        fft2048w<HS>(zre, zim, scr, scr_lds, twl, twh, lane);
        half_bwd(std::integral_constant<int, 0>{});
        // ---- odd output samples: zd = conj(A'[e]) D_lo[e] - A'[2048 - e] D_hi[e]
        {
            auto step = [&](auto cc) {                                    // four rows at a time (registers): k = 4 C4 .. 4 C4 + 3
                constexpr int C4 = decltype(cc)::value;
                unsigned vo = 2u * lane4;
                if constexpr (C4 > 0) asm volatile("" : "+v"(vo), "+v"(zre[4 * C4 - 1]), "+v"(zim[4 * C4 - 1]) : : "memory");
                else asm volatile("" : "+v"(vo) : : "memory");
                v2f dl[4], dh[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    using f4v = float __attribute__((ext_vector_type(4)));
                    const f4v d = tab_ld<f4v>(Rtab, 2u * vo, 16384 + 1024 * (4 * C4 + j));   // (D_lo, D_hi) of bin 64 k + lane
                    dl[j].x = d.x; dl[j].y = d.y; dh[j].x = d.z; dh[j].y = d.w;
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
                    zre[k] = fmaf(m[j].y, dh[j].y, fmaf(-m[j].x, dh[j].x, fmaf(a[j].y, dl[j].y, a[j].x * dl[j].x)));
                    zim[k] = fmaf(-m[j].y, dh[j].x, fmaf(-m[j].x, dh[j].y, fmaf(-a[j].y, dl[j].x, a[j].x * dl[j].y)));
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
        fft2048w<HS>(zre, zim, scr, scr_lds, twl, twh, lane);
        half_bwd(std::integral_constant<int, 1>{});
        float dpw = qacc / (half * half);
        // next task: reserved before the reductions
        const int tn = pull();
        int nset_i = 0, nrole = 0;
        if (tn < ntasks) decode(tn, nset_i, nrole);
        amu = wave_sum(amu);
        asg = wave_sum(asg);
        dpw = wave_sum(dpw);
        if (lane == 0) {
            const float sp = pool_sigma(p.pool_w[f], SKr);
            p.dkpart[((size_t)gb * p.F + f) * 2] = amu;
            p.dkpart[((size_t)gb * p.F + f) * 2 + 1] = asg;
            p.dwpart[(size_t)gb * p.F + f] = dpw / (sp * sp * sp);
        }
        if constexpr (DX) {
            if (role == p.F) {
                // the block's last filter in queue order: its second half was the last share (this wave added it)
                wg4k_dx_finish(p, gsum + slot * kWg4RingFloat2, gsum + 2 * kWg4RingFloat2, gb, ROT, lane, scr, scr_lds, twl, twh, tw4a,
                               tw4b);
                wg_release();                                             // cleared: the arrays' next blocks may add
                if (lane == 0) {
                    __hip_atomic_fetch_add(&gtick[slot], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    __hip_atomic_fetch_add(&gtick[2], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                }
            }
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

}  // namespace
