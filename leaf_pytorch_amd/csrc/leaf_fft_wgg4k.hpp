// leaf_fft_wgg4k.hpp -- the 4096-sample overlap-save forward (leaf_fft_wg4k.hpp) for ANY odd window 833 <= K <= 2049:
// window, hop and block length at run time.
// Part of the single translation unit leaf_kernels.hip (gfx950 only); see that file's header comment.
//
// With 2048-sample blocks a window of K = 1103 / 1201 (44.1 / 48 kHz) leaves 44 / 41 % valid outputs per transform, and
// K > 1217 does not fit at all; 4096-sample blocks leave 73 / 71 % and reach K = 2049.  Queue, ring, the decimation-in-time
// forward task and the two decimation-in-frequency half transforms per filter are those of leaf_fft_wg4k_kernel; the
// pooling is the run-time-geometry one of leaf_fft_wgg.hpp applied per half: half h holds the outputs n_c + 2 k + h, its
// |y|^2 go to a wave-private LDS row [K/2 zeros][2048][zeros], and a frame whose window starts at block sample is_m reads
// them from k0 = ceil((is_m - h) / 2) on with the taps of parity rho = (is_m - h) & 1 (g[rho], g[rho + 2], ...: the
// de-interleaved pooling rows of the 4096-sample tables), lane l holding taps l, 64 + l, ... of BOTH parities in registers.
// The frames of one parity class share a loop (hop even: one class per half; hop odd: the classes alternate), four frames
// per wave reduction; the first half parks its frame sums in a small wave-private LDS array, the second adds and stores.
#pragma once
#include "leaf_fft_wg4k.hpp"
#include "leaf_fft_wgg.hpp"

namespace {

// taps per lane and parity of the instantiation that serves window K: ceil(ceil(K / 2) / 64), bucketed
constexpr int fft_wgg4k_taps_per_lane(int K) { return K <= 1280 ? 10 : K <= 1664 ? 13 : 17; }
constexpr int kWgg4MaxFrames = 256;                                      // frames a block may meet (parked between the halves)
constexpr int fft_wgg4k_row_floats(int K) { return (kGPad + 64 * fft_wgg4k_taps_per_lane(K) + 3) / 4 * 4; }   // table row per parity
constexpr int fft_wgg4k_front_floats(int K) { return (K / 2 + 1 + 3) / 4 * 4; }
constexpr int fft_wgg4k_back_floats(int K) { return (64 * fft_wgg4k_taps_per_lane(K) - K / 2 + 3) / 4 * 4 + 4; }
// fbn = floats of the wave's frame-sum array (forward: the frames a block can meet, rounded up; backward: none)
constexpr int fft_wgg4k_frame_floats(int K, int hop) { return (((kFft4N - K + 1) + K - 2) / hop + 2 + 3) / 4 * 4; }
constexpr size_t fft_wgg4k_wave_floats(int K, int fbn) {
    return (size_t)fft_wgg4k_front_floats(K) + kFftN + fft_wgg4k_back_floats(K) + (size_t)fbn;
}
constexpr size_t fft_wgg4k_lds_bytes(int NW, int K, int fbn) {
    return ((size_t)kTwFloats + 2 * (32 + 64) + 2 * 2 * kWg4RingFloat2 + kWgQueueInts + (size_t)NW * fft_wgg4k_wave_floats(K, fbn)) * 4;
}

template <int NW, int NI2>
__global__ __launch_bounds__(NW * 64, (NW + 3) / 4) void leaf_fft_wgg4k_kernel(const FftParams p) {
    using gfp = const __attribute__((address_space(1))) float*;          // table pointers that stay `global` when made opaque
    using gf2p = const __attribute__((address_space(1))) v2f*;
    extern __shared__ __attribute__((aligned(16))) float wsm[];
    float2* twl = reinterpret_cast<float2*>(wsm);                        // [32][64]
    float2* twh = twl + 32 * 64;                                          // [32][2]
    float2* tw4a = twh + 64;                                              // w^(64 k), k < 32
    float2* tw4b = tw4a + 32;                                             // w^lane
    float2* ring = tw4b + 64;                                             // [2][kWg4RingFloat2]
    int* q = reinterpret_cast<int*>(ring + 2 * kWg4RingFloat2);
    const int PF = fft_wgg4k_front_floats(p.K), BP = fft_wgg4k_back_floats(p.K);
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane0 = tid & 63;
    float* wbase = reinterpret_cast<float*>(q + kWgQueueInts) + (size_t)wave * (PF + kFftN + BP + p.NT);   // p.NT: frame-sum floats
    float* scr = wbase + PF;                                              // energies [0, 2048); transposition scratch in its head
    float* fb = scr + kFftN + BP;                                         // frame sums of the first half
    const unsigned scr_lds = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(__attribute__((address_space(3))) float*)scr);

    fft_build_twiddles_wg(twl, twh, tid, (int)blockDim.x);
    for (int i = tid; i < 96; i += (int)blockDim.x) {
        float s, c;
        sincospif(2.0f * (float)(i < 32 ? 64 * i : i - 32) / (float)kFft4N, &s, &c);
        tw4a[i] = make_float2(c, -s);                                     // (tw4b follows tw4a contiguously)
    }
    if (tid < kWgQueueInts) q[tid] = 0;
    for (int i = lane0; i < PF; i += 64) wbase[i] = 0.0f;                 // written once: nothing else touches the paddings
    for (int i = lane0; i < BP; i += 64) scr[kFftN + i] = 0.0f;
    __syncthreads();

    const int PADL = p.padL, ROT = p.K / 2, LS = p.L, SKr = p.K, SHOPr = p.hop;   // odd windows: ROT = PADL

    // blocks dealt contiguously; clips all of whose blocks this workgroup ran are finalized in its tail (as leaf_fft_wg_kernel)
    const OwnedClips deal{p.B * p.nblk, (int)gridDim.x, p.nblk};
    const int first_gb = deal.start((int)blockIdx.x);
    const int nset = deal.count((int)blockIdx.x);
    const WgTaskGrid grid = wg_task_grid(p.F, nset);                       // F + 1 slots per set
    const int ntasks = nset > 0 ? 1 + nset * (p.F + 1) : 0;
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
        if (role == 0 || role > p.F) {
            if (role == 0 && set < nset) {
                // ---- A' = FFT4096(rotated block), bins 0..2048, by decimation in time: Xe = FFT2048(even samples) parked in
                // the ring slot, Xo = FFT2048(odd samples), A'[e] = Xe[e] + w^e Xo[e], A'[2048] = Xe[0] - Xo[0]
                const int gb = first_gb + set;
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
                fft2048w<true>(xre, xim, scr, scr_lds, twl, twh, lane);
                wg_wait_ge(&q[3 + slot], gen * p.F);                      // the slot's previous readers are done
#pragma unroll
                for (int i = 0; i < 32; ++i) A[64 * brev5(i) + lane] = make_float2(xre[i], xim[i]);
#pragma unroll
                for (int r = 0; r < 32; ++r) { xre[r] = sample(2 * (64 * r + lane) + 1); xim[r] = 0.0f; }
                fft2048w<true>(xre, xim, scr, scr_lds, twl, twh, lane);
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
        // ---- filter f of the block in ring slot `slot`
        const int f = role - 1;
        const float* Rtab = reinterpret_cast<const float*>(p.H) + (size_t)f * kFft4TabFloats;   // (R_lo, R_hi)[2048] f2 | (D_lo, D_hi)[2048] f4
        const unsigned lane4 = 4u * (unsigned)lane;
        if (set != seen_set) {                                            // this wave's first filter of the block: once the
            wg_wait_ge(&q[1 + slot], gen + 1);                            // spectrum is in the ring it stays until every filter is done
            seen_b = __builtin_amdgcn_readfirstlane(wg_ld(&q[5 + 2 * slot]));
            seen_c = __builtin_amdgcn_readfirstlane(wg_ld(&q[6 + 2 * slot]));
            seen_set = set;
        }
        const int b = seen_b, c = seen_c;
        const int n_c = c * LS;
        const int Lv = min(LS, p.T - n_c);
        int mlo = n_c + PADL - SKr + 1;
        mlo = mlo <= 0 ? 0 : (mlo + SHOPr - 1) / SHOPr;
        const int mhi = min(p.TP - 1, (n_c + Lv - 1 + PADL) / SHOPr);
        const unsigned a_dir = lds_addr(A + lane), a_mir = lds_addr(A + (2048 - 64 * 31) - lane);
        float zre[32], zim[32];
        // pooling of one half (FIRST: park the frame sums; else add the parked ones and store)
        auto pool_half = [&](auto hh) {
            constexpr int h = decltype(hh)::value;
            // the filter's pooling taps, both parities: lane l holds g[2 (64 i + l) + rho]  (table rows are zero past the window)
            float w0[NI2], w1[NI2];
            {
                gfp q0 = (gfp)(p.Gz + (size_t)f * 2 * p.GZ + kGPad) + lane;
                asm volatile("" : "+v"(q0) : : "memory");                // opaque (pointer, global address space): keeps the loads below the transform
                gfp q1 = q0 + p.GZ;
#pragma unroll
                for (int i = 0; i < NI2; ++i) {
                    w0[i] = q0[64 * i];
                    w1[i] = q1[64 * i];
                }
            }
            using lds_fp = __attribute__((address_space(3))) float*;
            using lds_cfp = const __attribute__((address_space(3))) float*;
            const lds_fp erow = (lds_fp)scr + lane;
#pragma unroll
            for (int i = 0; i < 32; ++i) {
                const int r = brev5(i);
                erow[64 * r] = 2 * (64 * r + lane) + h < Lv ? zre[i] * zre[i] + zim[i] * zim[i] : 0.0f;
            }
            // frame m: window start is_m = m hop - padL - n_c (block samples); first half-rate sample k0 = ceil((is_m - h) / 2),
            // tap parity rho = (is_m - h) & 1.  Frames of one parity class: all of them (hop even) or every other one.
            const int is_lo = mlo * SHOPr - PADL - n_c - h;
            const int step = (SHOPr & 1) ? 2 : 1;
            const lds_cfp ebase = (lds_cfp)scr + lane;
            auto run_class = [&](const float (&w)[NI2], int m_first) {
                if (m_first > mhi) return;
                const int m_last = m_first + (mhi - m_first) / step * step;      // last frame of the class
#pragma nounroll
                for (int m4 = m_first; m4 <= mhi; m4 += 4 * step) {
                    float a[4];
                    lds_cfp pk[4];
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const int m = min(m4 + k * step, m_last);
                        const int isx = m * SHOPr - PADL - n_c - h;              // is_m - h
                        pk[k] = ebase + ((isx + 1) >> 1);                        // ceil((is_m - h) / 2), arithmetic shift
                        a[k] = 0.0f;
                    }
#pragma unroll
                    for (int i = 0; i < NI2; ++i)
#pragma unroll
                        for (int k = 0; k < 4; ++k) a[k] = fmaf(w[i], pk[k][64 * i], a[k]);
                    float v = wave_sum4_rows(a[0], a[1], a[2], a[3]);     // 16-lane row q holds frame m4 + q step
                    const int m = m4 + (lane >> 4) * step;
                    if ((lane & 15) == 0 && m <= mhi) {
                        if (h == 0) {
                            fb[m - mlo] = v;
                        } else {
                            v += fb[m - mlo];
                            const int first = max(0, m * SHOPr - PADL);
                            int back = 0;
                            for (int nb = n_c; first < nb; nb -= LS) ++back;
                            p.part[(((size_t)b * p.F + f) * p.nslot + back) * p.TP + m] = v;
                        }
                    }
                }
            };
            const int rho_lo = is_lo & 1;                                  // parity class of frame mlo
            if (step == 1) {
                if (rho_lo) run_class(w1, mlo); else run_class(w0, mlo);
            } else {
                run_class(w0, mlo + (rho_lo ? 1 : 0));
                run_class(w1, mlo + (rho_lo ? 0 : 1));
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");            // row reads done before the next transform's scratch writes
        };
        // ---- even output samples: zs = conj(A'[e]) R_lo[e] + A'[2048 - e] R_hi[e]
        {
            auto chunk = [&](auto cc) {
                constexpr int C = decltype(cc)::value;
                float rl[8], rh[8];
                // the table offset is made opaque HERE: the loads below cannot issue before this point (a plain "memory"
                // clobber does not hold them -- they are hoisted under the previous phase and spilled one by one)
                unsigned vo = 2u * lane4;                                 // (tab_ld: uniform base + the lane's byte offset + an immediate)
                if constexpr (C > 0) asm volatile("" : "+v"(vo), "+v"(zre[8 * C - 1]), "+v"(zim[8 * C - 1]) : : "memory");
                else asm volatile("" : "+v"(vo) : : "memory");
#pragma unroll
                for (int j = 0; j < 8; ++j) { const v2f r = tab_ld<v2f>(Rtab, vo, 512 * (8 * C + j)); rl[j] = r.x; rh[j] = r.y; }
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
        fft2048w<true>(zre, zim, scr, scr_lds, twl, twh, lane);
        pin32(zre);
        pin32(zim);
        pool_half(std::integral_constant<int, 0>{});
        // the first half's pooling is complete before the second half's table loads issue
        asm volatile("" ::: "memory");
        // ---- odd output samples: zd = conj(A'[e]) D_lo[e] - A'[2048 - e] D_hi[e]
        {
            auto step = [&](auto cc) {                                    // four rows at a time (registers): k = 4 C4 .. 4 C4 + 3
                constexpr int C4 = decltype(cc)::value;
                using f4v = float __attribute__((ext_vector_type(4)));
                unsigned vo = 4u * lane4;
                if constexpr (C4 > 0) asm volatile("" : "+v"(vo), "+v"(zre[4 * C4 - 1]), "+v"(zim[4 * C4 - 1]) : : "memory");
                else asm volatile("" : "+v"(vo) : : "memory");
                v2f dl[4], dh[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const f4v d = tab_ld<f4v>(Rtab, vo, 16384 + 1024 * (4 * C4 + j));
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
        wg_release();
        if (lane == 0) __hip_atomic_fetch_add(&q[3 + slot], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);   // ring reads done
        fft2048w<true>(zre, zim, scr, scr_lds, twl, twh, lane);
        pin32(zre);
        pin32(zim);
        const int tn = pull();                                            // next task reserved under the pooling
        int nset_i = 0, nrole = 0;
        if (tn < ntasks) decode(tn, nset_i, nrole);
        pool_half(std::integral_constant<int, 1>{});
        t = tn;
        set = nset_i;
        role = nrole;
    }
    if (p.fin_fused)                                                      // the waves' rows are free: tile memory of the tail
        wg_tail_finalize<32>(p.fin, (first_gb + p.nblk - 1) / p.nblk, (first_gb + nset) / p.nblk,
                             reinterpret_cast<float*>(q + kWgQueueInts), tid, (int)blockDim.x);
}

}  // namespace
