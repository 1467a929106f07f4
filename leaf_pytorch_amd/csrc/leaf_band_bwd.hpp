// leaf_band_bwd.hpp -- band-limited filter tasks of the static workgroup BACKWARD kernel (leaf_fft_wg_bwd.hpp), round 5
// Part of libleaf_hip.so (gfx950 only); included by leaf_fft_wg_bwd.hpp.
//
// The parameter gradients of the filters the forward runs on 256- / 512-point transforms (leaf_band.hpp), at the same decimated
// rate.  With Zb[j] = conj(A'[kb + j]) R[j] the window's bins, z = F_M Zb (the band task's network: a plain M-point DFT matrix
// F, symmetric), e[q] = |z[q]|^2 and p[m] = sum_q W_m[q] e[q] (W_m: the decimated pooling window G~ of a regular frame, the dense
// edge table of an edge frame):
//     de[q]   = sum_m g_pre[m] W_m[q]                      (pooling backward: the forward's weights, transposed)
//     v[q]    = 2 de[q] conj(z[q]),   V = F_M v            (the SAME network once more, after a re-layout through the wave's scratch)
//     dL/dR[j] = Re(conj(A'[kb + j]) V[j])                 -> d mu, d sigma as dot products with the R_mu, R_sigma tables on the window
//     d pool_w = sum_q e[q] sum_m g_pre[m] W2_m[q]         W2: the same tables built from g[j] (j - c)^2 (fft_prep_band_kernel: gz2, edge2)
// (the time-shift phase that z lacks against y_f cancels in conj(z) z-terms: it multiplies z[q] and divides v[q]).  What the
// window drops of R's derivatives is of the order of what it drops of R: the gradients agree with the full-transform backward to
// ~1e-5 of their largest component (tests: 1e-4 against fp64 autograd through the oracle).  Eight / four filters per task as in
// the forward; per (block, filter) partials into dkpart / dwpart like the full task (no atomics: bit-reproducible).
#pragma once
#include "leaf_fft_wg.hpp"

namespace {

// The M-point network of a band task behind the first decimation-in-time stage of the 16-point transforms over j2 (the forward
// fuses that stage with the spectral multiply): phase-1 layout in (lane = (filter, column), register (h, r) <-> index
// c + (A/2) h + A r), phase-2 layout out (lane = (m2, filter), register <-> m1: index 16 m1 + m2).  band_task's own code.
template <int A>
__device__ __forceinline__ void band_network(float (&zre)[32], float (&zim)[32], float (&tr)[32], float (&ti)[32], const float2* twl,
                                             float* scr, unsigned scr_lds, int lane) {
    constexpr int LPF = band_lpf(A), G = band_d(A);
    const int c1 = lane & (LPF - 1);
    band_dit16_stage<2, 0>(zre, zim);
    band_dit16_stage<2, 16>(zre, zim);
    band_dit16_stage<4, 0>(zre, zim);
    band_dit16_stage<4, 16>(zre, zim);
    band_dit16_stage<8, 0>(zre, zim);
    band_dit16_stage<8, 16>(zre, zim);
    lds_stream32(lds_addr(twl + (64 / A) * c1), OffBandTw{}, [&](int k, v2f w) {
        if (brev4(k & 15) == 0) return;
        const float r = zre[k] * w.x - zim[k] * w.y;
        zim[k] = zre[k] * w.y + zim[k] * w.x;
        zre[k] = r;
    });
    const int g2 = lane & (G - 1), l2 = lane / G;
    {
        const f32x4* row = reinterpret_cast<const f32x4*>(scr + l2 * kWgScrStride + g2 * LPF);
        constexpr int R16 = 16 * kWgScrStride / 4, R8 = 8 * kWgScrStride / 4;
        auto plane = [&](const float (&src)[32], float (&t)[32]) {
            band_transpose_store(src, scr_lds);
            f32x4 v[8];
            if constexpr (A == 16) {
                v[0] = row[0]; v[1] = row[1]; v[2] = row[R16]; v[3] = row[R16 + 1];
                v[4] = row[R8]; v[5] = row[R8 + 1]; v[6] = row[R8 + R16]; v[7] = row[R8 + R16 + 1];
            } else {
                v[0] = row[0]; v[1] = row[1]; v[2] = row[2]; v[3] = row[3];
                v[4] = row[R16]; v[5] = row[R16 + 1]; v[6] = row[R16 + 2]; v[7] = row[R16 + 3];
            }
#pragma unroll
            for (int q = 0; q < 8; ++q) { t[4 * q] = v[q].x; t[4 * q + 1] = v[q].y; t[4 * q + 2] = v[q].z; t[4 * q + 3] = v[q].w; }
            asm volatile("" ::: "memory");
        };
        pin32(zre);
        pin32(zim);
        plane(zre, tr);
        pin32(tr);
        plane(zim, ti);
        pin32(ti);
    }
    if constexpr (A == 16) {
        band_dit16_stage<1, 0>(tr, ti);
        band_dit16_stage<1, 16>(tr, ti);
        band_dit16_stage<2, 0>(tr, ti);
        band_dit16_stage<2, 16>(tr, ti);
        band_dit16_stage<4, 0>(tr, ti);
        band_dit16_stage<4, 16>(tr, ti);
        band_dit16_stage<8, 0>(tr, ti);
        band_dit16_stage<8, 16>(tr, ti);
    } else {
        fft32_dif(tr, ti);
    }
}

// index (decimated sample / bin offset in the window) of register k in the phase-2 layout, without the lane's m2 = l2 part
template <int A>
__device__ __forceinline__ constexpr int band_p2_index(int k) {
    return A == 32 ? 16 * brev5(k) : 16 * brev4(k & 15) + 8 * (k >> 4);
}

// One band task of the backward.  rq, Aring, mem, elist, twl, scr: as band_task.  Writes the (d mu, d sigma) and d pool_w
// partials of its member filters for block gb; the caller counts the task as ONE reader of the ring slot.
// DXB (2048-sample plan, dL/dx): the members' shares R V of the block's folded gradient spectrum gS (leaf_fft_wg_bwd.hpp: with
// g the full task's gradient spectrum at the filter's entries 2048 - bin, V[j] = conj(g): the share conj(R g) of bin kb + j is R V)
// are added member after member, plain read-add-write, in the task's turn of the block's order (gticket == want; the caller's
// tasks take their turns in queue order): no float atomics, the sum order does not depend on timing.  Before its turn the task
// spreads every member's window over all 64 lanes through its scratch, so that the turn -- one link of the block's chain of
// turns, which bounds the kernel when it is long (profiles/r05/ab_band_dx.txt) -- is M / 64 rows per member.
template <int A, int SK, int SHOP, bool N4K = false, bool DXB = false>
__device__ __forceinline__ void band_bwd_task(const FftParams& p, const float (&rq)[32], const float2* Aring, const int* mem, const int* elist,
                                              const float2* twl, float* scr, unsigned scr_lds, int b, int c, int gb, int mlo, int mhi, int lane,
                                              [[maybe_unused]] float2* gS = nullptr, [[maybe_unused]] int* gticket = nullptr,
                                              [[maybe_unused]] int want = 0) {
    using GEO = BandGeom<A, SK, SHOP, N4K>;   // N4K: the 4096-sample plan of the 32 kHz window (512-bin window of the 4096-point spectrum, decimation 8)
    constexpr int LPF = band_lpf(A), D = GEO::D, G = band_d(A), RL = GEO::RL, M = band_m(A);
    constexpr int PADL = SK / 2 + SK % 2 - 1, LS = GEO::LS;
    constexpr int DMIN = -((SK - 1 - PADL) / SHOP), DMAX = (LS - 1 + PADL) / SHOP, NFR = DMAX - DMIN + 1;
    constexpr int LPHI = kBandLh * D, PG = band_gcd(RL, SHOP), C0MIN = band_c0min_d(SK, SHOP, A, D), NV = band_nv_d(SK, SHOP, A, D);
    constexpr int MP = M + (A == 32 ? 8 : 4);                            // per-filter stride of the re-layout (bank spread)
    static_assert((N4K || band_geometry_ok(SK, SHOP)) && NFR <= 16 && G * MP <= kWgScrFloats, "band tasks: static geometry");
    const int g2 = lane & (G - 1), l2 = lane / G;                         // phase-2 lane: (m2, filter)
    const int me2 = mem[g2];
    const int fid2 = me2 & 0xffff, kb2 = (me2 >> 16) & 0x7ff;
    const bool valid = !(me2 & kBandInvalid);
    const int n_c = c * LS;
    float zre[32], zim[32];
    {
        // Z = conj(A'[k]) R fused with the first stage (band_task)
        const int me1 = mem[lane / LPF];
        const int kb = (me1 >> 16) & 0x7ff, c1 = lane & (LPF - 1);
        const unsigned a0 = lds_addr(Aring + kb + c1);
        v2f av[4][8];
        constexpr auto seq = std::make_integer_sequence<int, 8>{};
        auto pairs = [&](auto hh) {
            constexpr int h = decltype(hh)::value;
#pragma unroll
            for (int r = 0; r < 8; ++r) {
                const float ra = rq[16 * h + r], rb = rq[16 * h + r + 8];
                const v2f xa = av[2 * h][r], xb = av[2 * h + 1][r];
                const float tr_ = xa.x * ra, ti_ = -(xa.y * ra);
                zre[16 * h + r] = fmaf(xb.x, rb, tr_);
                zim[16 * h + r] = fmaf(-xb.y, rb, ti_);
                zre[16 * h + r + 8] = fmaf(-xb.x, rb, tr_);
                zim[16 * h + r + 8] = fmaf(xb.y, rb, ti_);
            }
        };
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        band_rd_chunk<A, 0>(av[0], a0, seq);
        band_rd_chunk<A, 1>(av[1], a0, seq);
        band_rd_chunk<A, 2>(av[2], a0, seq);
        lds_wait8<8>(av[0]);
        lds_wait8<8>(av[1]);
        pairs(std::integral_constant<int, 0>{});
        band_rd_chunk<A, 3>(av[3], a0, seq);
        lds_wait8<0>(av[2]);
        lds_wait8<0>(av[3]);
        pairs(std::integral_constant<int, 1>{});
    }
    float tr[32], ti[32];
    band_network<A>(zre, zim, tr, ti, twl, scr, scr_lds, lane);           // z in the phase-2 layout
    // g_pre of this lane's filter at the block's frames: regular frames as NFR values (requested behind the first network: thirteen registers less across it)
    const float* gprow = p.gpre + ((size_t)b * p.F + fid2) * p.TP;
    float gp[NFR];
    {
        const int rlo = max(mlo, p.band.reg_lo), rhi = min(mhi, p.band.reg_hi);
        asm volatile("" ::: "memory");
#pragma unroll
        for (int fi = 0; fi < NFR; ++fi) {
            const int m = n_c / SHOP + DMIN + fi;
            const bool on = valid && m >= rlo && m <= rhi;
            const float v = gprow[min(max(m, 0), p.TP - 1)];
            gp[fi] = on ? v : 0.0f;
        }
        asm volatile("" ::: "memory");
    }
    // ---- pooling backward at the decimated rate: 2 de[k] with the forward's weights G~, then the d pool_w share with G~2 (one
    // weight set in registers at a time)
    float s2[32], dpw = 0.0f;
#pragma unroll
    for (int k = 0; k < 32; ++k) s2[k] = 0.0f;
    {
        float pw[NV];
        const float* gsrc = p.band.gz + (size_t)fid2 * GEO::GZF + GEO::GZ0 + l2;
        asm volatile("" ::: "memory");
#pragma unroll
        for (int k = 0; k < NV; ++k) pw[k] = gsrc[PG / D * k];
        asm volatile("" ::: "memory");
#pragma unroll
        for (int rho = 0; rho < LS / RL; ++rho) {
            const int k = A == 32 ? brev5(rho) : 16 * (rho & 1) + brev4(rho >> 1);   // register of row rho (band_task)
            float de = 0.0f;
#pragma unroll
            for (int fi = 0; fi < NFR; ++fi) {
                const int c0 = RL * rho - ((DMIN + fi) * SHOP - PADL);
                if (c0 >= C0MIN && c0 <= SK - 1 + LPHI) de = fmaf(gp[fi], pw[(c0 - C0MIN) / PG], de);
            }
            s2[k] = 2.0f * de;
        }
    }
    {
        float pw2[NV];
        const float* gsrc2 = p.band.gz2 + (size_t)fid2 * GEO::GZF + GEO::GZ0 + l2;
        asm volatile("" : "+v"(s2[0]) : : "memory");
#pragma unroll
        for (int k = 0; k < NV; ++k) pw2[k] = gsrc2[PG / D * k];
        asm volatile("" ::: "memory");
#pragma unroll
        for (int rho = 0; rho < LS / RL; ++rho) {
            const int k = A == 32 ? brev5(rho) : 16 * (rho & 1) + brev4(rho >> 1);
            float dq = 0.0f;
#pragma unroll
            for (int fi = 0; fi < NFR; ++fi) {
                const int c0 = RL * rho - ((DMIN + fi) * SHOP - PADL);
                if (c0 >= C0MIN && c0 <= SK - 1 + LPHI) dq = fmaf(gp[fi], pw2[(c0 - C0MIN) / PG], dq);
            }
            dpw = fmaf(tr[k] * tr[k] + ti[k] * ti[k], dq, dpw);
        }
    }
    // edge frames of this block (its first and the clip's last blocks only): dense tables over all 32 registers
    if (c == 0 || c >= p.nblk - 2) {
        const int n_edge = p.band.n_edge;
        const size_t eoff = ((size_t)fid2 * GEO::NCLS + GEO::CLS) * kBandMaxEdge * 512 + l2;
        for (int s = 0; s < n_edge; ++s) {
            if (__builtin_amdgcn_readfirstlane(elist[4 * s]) != c) continue;
            const int m = __builtin_amdgcn_readfirstlane(elist[4 * s + 1]);
            const float gs = valid ? gprow[m] : 0.0f;
            const float* t1 = p.band.edge + eoff + (size_t)s * 512;
            const float* t2 = p.band.edge2 + eoff + (size_t)s * 512;
#pragma unroll
            for (int k = 0; k < 32; ++k) {
                s2[k] = fmaf(2.0f * gs, t1[k * LPF], s2[k]);
                dpw = fmaf((tr[k] * tr[k] + ti[k] * ti[k]) * gs, t2[k * LPF], dpw);
            }
        }
    }
#pragma unroll
    for (int k = 0; k < 32; ++k) {
        tr[k] = s2[k] * tr[k];                                            // v = 2 de conj(z)
        ti[k] = -(s2[k] * ti[k]);
    }
    // ---- v from the phase-2 layout into the phase-1 layout through the wave's scratch: [filter][MP], one plane at a time
    {
        float* wr = scr + g2 * MP + l2;
        const int g1 = lane / LPF, c1 = lane & (LPF - 1);
        const float* rd = scr + g1 * MP + c1;
        auto relayout = [&](const float (&src)[32], float (&dst)[32]) {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
            for (int k = 0; k < 32; ++k) wr[band_p2_index<A>(k)] = src[k];
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
            for (int k = 0; k < 32; ++k) dst[k] = rd[(A / 2) * (k >> 4) + A * (k & 15)];
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        };
        relayout(tr, zre);
        relayout(ti, zim);
    }
    band_dit16_stage<1, 0>(zre, zim);                                     // the stage the forward fuses with its multiply
    band_dit16_stage<1, 16>(zre, zim);
    band_network<A>(zre, zim, tr, ti, twl, scr, scr_lds, lane);           // V[j], j = 16 m1 + m2, in the phase-2 layout
    // ---- dL/dR[j] = Re(conj(A'[kb + j]) V[j]) against the derivative tables on the window's bins (entries 2048 - bin)
    float amu = 0.0f, asg = 0.0f;
    if constexpr (N4K) {
        // (4096-sample plan: the derivative tables hold (d/dmu lo, hi, d/dsigma lo, hi)[e] as one 16-byte entry behind the mu slab's
        // R table -- fft4k_prep_kernel; bin k of the positive half is entry 4096 - k = "hi" of e = 2048 - k)
        using f4v = float __attribute__((ext_vector_type(4)));
        const float2* ap = Aring + kb2 + l2;
        const f4v* dt = reinterpret_cast<const f4v*>(reinterpret_cast<const float*>(p.H) + ((size_t)p.F + fid2) * 12288 + 4096) + (2048 - kb2 - l2);
#pragma unroll
        for (int k0 = 0; k0 < 32; k0 += 8) {
            f4v tv[8];
            float2 av[8];
            asm volatile("" : "+v"(amu), "+v"(asg) : : "memory");
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int off = band_p2_index<A>(k0 + j);
                tv[j] = dt[-off];
                av[j] = ap[off];
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float d = av[j].x * tr[k0 + j] + av[j].y * ti[k0 + j];
                amu = fmaf(d, tv[j].y, amu);
                asg = fmaf(d, tv[j].w, asg);
            }
        }
    } else {
        const float2* ap = Aring + kb2 + l2;
        const float* rmu = reinterpret_cast<const float*>(p.H) + ((size_t)p.F + fid2) * kFftN + (kFftN - kb2 - l2);
        const float* rsg = reinterpret_cast<const float*>(p.H) + ((size_t)2 * p.F + fid2) * kFftN + (kFftN - kb2 - l2);
#pragma unroll
        for (int k0 = 0; k0 < 32; k0 += 8) {
            float tm[8], ts[8];
            float2 av[8];
            asm volatile("" : "+v"(amu), "+v"(asg) : : "memory");
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int off = band_p2_index<A>(k0 + j);
                tm[j] = rmu[-off];
                ts[j] = rsg[-off];
                av[j] = ap[off];
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float d = av[j].x * tr[k0 + j] + av[j].y * ti[k0 + j];
                amu = fmaf(d, tm[j], amu);
                asg = fmaf(d, ts[j], asg);
            }
        }
    }
    if constexpr (DXB && !N4K) {
        // R at this lane's bins (the values rq held in the phase-1 layout), then the turn
        const float* rr = reinterpret_cast<const float*>(p.H) + (size_t)fid2 * kFftN + (kFftN - kb2 - l2);
#pragma unroll
        for (int k0 = 0; k0 < 32; k0 += 8) {
            float rv[8];
            asm volatile("" : "+v"(tr[k0]), "+v"(ti[k0]) : : "memory");
#pragma unroll
            for (int j = 0; j < 8; ++j) rv[j] = rr[-band_p2_index<A>(k0 + j)];
#pragma unroll
            for (int j = 0; j < 8; ++j) { tr[k0 + j] *= rv[j]; ti[k0 + j] *= rv[j]; }
        }
        // member g's window across all 64 lanes (through the wave's scratch, one plane at a time): register g (M / 64) + j <-> bin
        // kb_g + 64 j + lane -- the turn below then costs one short read-add-write per member instead of one per member on an eighth
        // (a quarter) of the lanes
        constexpr int RPM = M / 64;                                       // registers per member
        {
            float* wr = scr + g2 * MP + l2;
            const float* rd = scr + lane;
            auto spread = [&](float (&v)[32]) {
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
                for (int k = 0; k < 32; ++k) wr[band_p2_index<A>(k)] = v[k];
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
                for (int k = 0; k < 32; ++k) v[k] = rd[(k / RPM) * MP + 64 * (k % RPM)];
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            };
            spread(tr);
            spread(ti);
        }
#ifndef LEAF_DX_NOWAIT                 // measurement only (wrong sums): what the ordered turn costs
        wg_wait_ge(gticket, want);
#endif
        if (LEAF_DX_PRIO) __builtin_amdgcn_s_setprio(3);
#pragma unroll
        for (int g = 0; g < G; ++g) {                                     // member after member: their windows overlap (a wave's LDS
            const int me = __builtin_amdgcn_readfirstlane(mem[g]);        // operations execute in order)
            if (me & kBandInvalid) continue;
            float2* sp = gS + ((me >> 16) & 0x7ff) + lane;
            float2 sv[RPM];
#pragma unroll
            for (int j = 0; j < RPM; ++j) sv[j] = sp[64 * j];
#pragma unroll
            for (int j = 0; j < RPM; ++j) {
                sv[j].x += tr[g * RPM + j];
                sv[j].y += ti[g * RPM + j];
                sp[64 * j] = sv[j];
            }
        }
        wg_release();
        if (lane == 0) __hip_atomic_fetch_add(gticket, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        if (LEAF_DX_PRIO) __builtin_amdgcn_s_setprio(0);
    }
    amu = band_filter_sum<A>(amu);
    asg = band_filter_sum<A>(asg);
    dpw = band_filter_sum<A>(dpw);
    if (valid && l2 == 0) {
        constexpr float HALFW = 0.5f * (float)(SK - 1);
        const float sp = pool_sigma(p.pool_w[fid2], SK);
        p.dkpart[((size_t)gb * p.F + fid2) * 2] = amu;
        p.dkpart[((size_t)gb * p.F + fid2) * 2 + 1] = asg;
        p.dwpart[(size_t)gb * p.F + fid2] = dpw * (1.0f / (HALFW * HALFW)) / (sp * sp * sp);
    }
}

}  // namespace
