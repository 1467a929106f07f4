// leaf_fft_wgg_bwd.hpp -- the workgroup-per-block overlap-save BACKWARD (leaf_fft_wg_bwd.hpp) for ANY window the
// 2048-sample plan covers: window, hop and block length at run time, odd and even windows.
// Part of the single translation unit leaf_kernels.hip (gfx950 only); see that file's header comment.
//
// Same queue, ring, transforms and spectral tail as leaf_fft_wg_bwd_kernel; the pooling backward cannot unroll over
// compile-time frame offsets, so it uses the layout of the run-time-geometry forward (leaf_fft_wgg.hpp): a wave-private LDS
// row [K - 1 zeros][2048][zeros] and the filter's pooling taps in registers (lane l: taps l, 64 + l, ...).  Per (block, filter):
//   1. |y|^2 -> the row;  d pool_w:  sum_m g_pre[m] sum_j g_f[j] (j - c)^2 e[is_m + j]  -- the forward's gather with the taps
//      pre-multiplied by (j - c)^2, four frames per turn, no reduction until the end of the task;
//   2. the row is cleared and de[n] = sum_m g_pre[m] g_f[n - is_m] is built by scatter: per frame NI read-add-write triples
//      at immediate offsets from the frame's address (same-wave LDS operations execute in order, so overlapping windows of
//      consecutive frames need no fence beyond the data dependence);
//   3. de comes back row by row, gy = 2 de y, second transform, spectral dot products (wg_bwd_tail).
// Even windows: the unpaired tap t = -K/2 is added to y in the time domain and its gradient is two more sums over the
// block (see leaf_fft.hpp, RS = 2).
#pragma once
#include "leaf_fft_wg_bwd.hpp"
#include "leaf_fft_wgg.hpp"

namespace {

// Backward of ONE filter on ONE block whose spectrum A' (bins 0..1024) sits in LDS at A; rq = R_f[64 k + lane]; wbase = the
// wave's LDS row [K - 1 zeros][2048, the transposition scratch in its head (scr)][zeros].  Returns this lane's shares of
// d mu, d sigma, d pool_w (before the wave sums); DX = 2 also adds R_f g to the block's shared G (gS, in filter order by
// gticket).  `even`: the window has an unpaired tap; its share of dL/dx, Re(conj(c) gy[n]) per block sample, is summed the
// same way in tS[2048].  (acc_re, acc_im) are unused scratch references kept for wg_bwd_tail's signature.)
template <int NI, int DX, bool HALF = true>
__device__ __forceinline__ void wgg_bwd_filter(const FftParams& p, const float2* A, int lane, int f, int b, int c, bool even,
                                               const float (&rq)[32], float* wbase, float* scr, unsigned scr_lds, const float2* twl,
                                               const float2* twh, float (&acc_re)[32], float (&acc_im)[32], float& amu_out,
                                               float& asg_out, float& dpw_out,
                                               [[maybe_unused]] float2* gS = nullptr, [[maybe_unused]] const int* gticket = nullptr,
                                               [[maybe_unused]] int want = 0, [[maybe_unused]] float* tS = nullptr,
                                               [[maybe_unused]] const int* tticket = nullptr) {
    const int PF = fft_wgg_front_floats(p.K), BP = HALF ? fft_wgg_back_floats(p.K) : fft_wgg_back_floats_full(p.K);
    const int PADL = p.padL, LS = p.L, SKr = p.K, SHOPr = p.hop;
    using lds_fp = __attribute__((address_space(3))) float*;
    using f4 = float __attribute__((ext_vector_type(4)));
    using lds_f4p = __attribute__((address_space(3))) f4*;
    const f4 zero4 = {0.0f, 0.0f, 0.0f, 0.0f};
    const int n_c = c * LS;
    const int Lv = min(LS, p.T - n_c);
    int mlo = n_c + PADL - SKr + 1;                                   // first frame whose window reaches the block
    mlo = mlo <= 0 ? 0 : (mlo + SHOPr - 1) / SHOPr;
    const int mhi = min(p.TP - 1, (n_c + Lv - 1 + PADL) / SHOPr);
    float zre[32], zim[32];
    wg_ring_rows(A, lane, [&](int k, float ar, float ai) {           // Z = conj(A' R_f), natural row order
        zre[k] = ar * rq[k];
        zim[k] = -(ai * rq[k]);
    });
    fft2048w<HALF>(zre, zim, scr, scr_lds, twl, twh, lane);          // u = conj(y): register i <-> samples 64 brev5(i) + lane
    const int nb_x = n_c - PADL;                                      // (even windows) clip sample under the block's first sample
    const bool x_interior = nb_x >= 0 && nb_x + kFftN <= p.T;
    if (even) {
        // the unpaired tap t = -K/2: u += conj(c) x[n_c - padL + n]
        const float cre = p.lone[2 * f], cim = p.lone[2 * f + 1];
        pin32(zre);
        pin32(zim);
#pragma unroll
        for (int i0 = 0; i0 < 32; i0 += 8) {
            float xa[8];
            block_rows8(p.x, (size_t)b * p.T, 0, nb_x, p.T, lane, x_interior, [&](int j) { return brev5(i0 + j); }, xa);
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                zre[i0 + j] = fmaf(cre, xa[j], zre[i0 + j]);
                zim[i0 + j] = fmaf(-cim, xa[j], zim[i0 + j]);
            }
            asm volatile("" : "+v"(zre[i0]), "+v"(zre[i0 + 1]), "+v"(zre[i0 + 2]), "+v"(zre[i0 + 3]), "+v"(zre[i0 + 4]),
                              "+v"(zre[i0 + 5]), "+v"(zre[i0 + 6]), "+v"(zre[i0 + 7]), "+v"(zim[i0]), "+v"(zim[i0 + 1]),
                              "+v"(zim[i0 + 2]), "+v"(zim[i0 + 3]), "+v"(zim[i0 + 4]), "+v"(zim[i0 + 5]), "+v"(zim[i0 + 6]),
                              "+v"(zim[i0 + 7]));
        }
    }
    pin32(zre);
    pin32(zim);
    // the filter's pooling taps: lane l holds g_f[l], g_f[64 + l], ...; w2 = the same taps times (j - centre)^2
    float w[NI], w2[NI];
    {
        const float* gsrc = p.Gz + (size_t)f * p.GZ + kGPad;
        int ofs = 0;
        asm volatile("" : "+v"(ofs) : : "memory");
#pragma unroll
        for (int i = 0; i < NI; ++i) w[i] = gsrc[min(64 * i + lane, p.GZ - kGPad - 1) + ofs];
    }
    const lds_fp erow = (lds_fp)scr + lane;
    const lds_f4p zrow = (lds_f4p)wbase + lane;                       // 16 bytes per lane per store when clearing
    // 1. |y|^2 -> the row (zero where the block has no sample); the front and back paddings are cleared too -- the
    //    previous task's scatter ran into them
    for (int i0 = 0; i0 < PF; i0 += 256)
        if (i0 + 4 * lane < PF) zrow[i0 / 4] = zero4;
    for (int i0 = 0; i0 < BP; i0 += 256)
        if (i0 + 4 * lane < BP) zrow[(PF + kFftN + i0) / 4] = zero4;
#pragma unroll
    for (int i = 0; i < 32; ++i) {
        const int r = brev5(i);
        erow[64 * r] = 64 * r + lane < Lv ? zre[i] * zre[i] + zim[i] * zim[i] : 0.0f;
    }
    const float half = 0.5f * (float)(SKr - 1);
    {
        float tj = (float)lane - half;
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            w2[i] = w[i] * (tj * tj);
            tj += 64.0f;
        }
    }
    using lds_cfp = const __attribute__((address_space(3))) float*;
    const float* gp_row = p.gpre + ((size_t)b * p.F + f) * p.TP;      // g_pre of this (clip, filter)
    const int is0 = -PADL - n_c;                                      // window start of frame m relative to the block: m hop + is0
    float qacc = 0.0f;
    // d pool_w: gather, four frames per turn; frames past mhi repeat frame mhi with g_pre = 0
    {
        const lds_cfp ebase = (lds_cfp)scr + lane;
#pragma nounroll
        for (int mc = mlo; mc <= mhi; mc += 64) {                     // g_pre of up to 64 frames: one per lane
            const float mine = mc + lane <= mhi ? gp_row[mc + lane] : 0.0f;
            const int ncur = min(64, mhi - mc + 1);
#pragma nounroll
            for (int j4 = 0; j4 < ncur; j4 += 4) {
                float a[4];
                lds_cfp pk[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    pk[k] = ebase + (min(mc + j4 + k, mhi) * SHOPr + is0);
                    a[k] = 0.0f;
                }
#pragma unroll
                for (int i = 0; i < NI; ++i)
#pragma unroll
                    for (int k = 0; k < 4; ++k) a[k] = fmaf(w2[i], pk[k][64 * i], a[k]);
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const float gpk = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(mine), min(j4 + k, 63)));
                    qacc = fmaf(j4 + k < ncur ? gpk : 0.0f, a[k], qacc);
                }
            }
        }
    }
    // 2. de[n] = sum_m g_pre[m] g_f[n - is_m]: clear the row, then scatter (read, add, write).  Frames S = ceil(64 NI / hop)
    //    apart touch disjoint addresses, so the frames of one residue class mod S go BK at a time (BK NI reads in
    //    flight, BK sized to the registers); classes follow each other in program order -- the LDS executes a wave's operations in order, so a later
    //    class sees the earlier ones' writes.
    for (int i0 = 0; i0 < kFftN; i0 += 256) zrow[(PF + i0) / 4] = zero4;
    {
        const lds_fp dbase = (lds_fp)scr + lane;
        const int S = (64 * NI + SHOPr - 1) / SHOPr;
        constexpr int BK = NI <= 7 ? 4 : NI <= 10 ? 3 : NI <= 13 ? 2 : 1;
#pragma nounroll
        for (int mc = mlo; mc <= mhi; mc += 64) {
            const float mine = mc + lane <= mhi ? gp_row[mc + lane] : 0.0f;
            const int ncur = min(64, mhi - mc + 1);
#pragma nounroll
            for (int rho = 0; rho < min(S, ncur); ++rho) {
#pragma nounroll
                for (int j = rho; j < ncur; j += BK * S) {
                    float tv[BK][NI];
                    lds_fp pk[BK];
                    float gpm[BK];
#pragma unroll
                    for (int k = 0; k < BK; ++k) {
                        const int jk = j + k * S;                     // wave-uniform
                        pk[k] = dbase + ((mc + jk) * SHOPr + is0);
                        gpm[k] = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(mine), min(jk, 63)));
                        if (jk < ncur) {
#pragma unroll
                            for (int i = 0; i < NI; ++i) tv[k][i] = pk[k][64 * i];
                        }
                    }
#pragma unroll
                    for (int k = 0; k < BK; ++k) {
                        if (j + k * S < ncur) {
#pragma unroll
                            for (int i = 0; i < NI; ++i) pk[k][64 * i] = fmaf(gpm[k], w[i], tv[k][i]);
                        }
                    }
                }
            }
        }
    }
    // 3. gy = 2 de y (natural row order; no gradient past the clip's end or in the circular wrap-around rows)
    float vre[32], vim[32];
#pragma unroll
    for (int r = 0; r < 32; ++r) {
        const int i = brev5(r);
        const float de = erow[64 * r];
        const float s2 = 64 * r + lane < Lv ? 2.0f * de : 0.0f;
        vre[r] = s2 * zre[i];
        vim[r] = -(s2 * zim[i]);
    }
    float amu = 0.0f, asg = 0.0f;
    if (even) {
        // u = u_H + conj(c) x  =>  dL/dc_re = sum_n x[n] Re v[n], dL/dc_im = sum_n x[n] Im v[n]
        float lgr = 0.0f, lgi = 0.0f;
        pin32(vre);
        pin32(vim);
#pragma unroll
        for (int r0 = 0; r0 < 32; r0 += 8) {
            float xa[8];
            block_rows8(p.x, (size_t)b * p.T, 0, nb_x, p.T, lane, x_interior, [&](int j) { return r0 + j; }, xa);
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                lgr = fmaf(xa[j], vre[r0 + j], lgr);
                lgi = fmaf(xa[j], vim[r0 + j], lgi);
            }
            asm volatile("" : "+v"(lgr), "+v"(lgi));
        }
        const float* dmu = p.lone + ((size_t)p.F + f) * 2;
        const float* dsg = p.lone + ((size_t)2 * p.F + f) * 2;
        amu = dmu[0] * lgr + dmu[1] * lgi;
        asg = dsg[0] * lgr + dsg[1] * lgi;
        if constexpr (DX == 2) {
            // the unpaired tap's share of dL/dx, Re(gy[n] conj(c)) at the un-rotated block sample n = 64 r + lane, summed over
            // the block's filters in a second shared LDS array (tS, 2048 floats per ring slot), in filter order like G below
            const float cre = p.lone[2 * f], cim = p.lone[2 * f + 1];
            wg_wait_ge(tticket, want);
            float* t1 = tS + lane;
#pragma unroll
            for (int r0 = 0; r0 < 32; r0 += 8) {
                float old[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) old[j] = t1[64 * (r0 + j)];
#pragma unroll
                for (int j = 0; j < 8; ++j) t1[64 * (r0 + j)] = fmaf(cre, vre[r0 + j], fmaf(cim, vim[r0 + j], old[j]));
            }
            wg_release();
            if (lane == 0) __hip_atomic_fetch_add(const_cast<int*>(tticket), 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                // the row's reads are done before the transform's scratch writes
    fft2048w<HALF>(vre, vim, scr, scr_lds, twl, twh, lane);          // g = dL/dS: register i <-> bin 64 brev5(i) + lane
    pin32(vre);
    pin32(vim);
    wg_bwd_tail<0>(p, A, lane, f, vre, vim, acc_re, acc_im, amu, asg);
    if constexpr (DX == 2) wg_dx_accumulate(p, f, lane, vre, vim, gS, gticket, want);
    amu_out = amu;
    asg_out = asg;
    dpw_out = qacc / (half * half);
}

// DX = true: the kernel also yields dL/dx.  dL/dA'[k] = sum_f R_f[k] g_f[k] =: G[k] is accumulated per BLOCK in LDS
// (one Hermitian-folded array per ring slot, see wgg_bwd_filter<.., 2>) by the waves that run the block's filters, in filter
// order; the wave that adds the last filter turns it into the block's 2048 input-gradient samples (one more transform) and
// stores them, un-rotated, into part[block][2048]; fft_dx_gather_kernel sums the overlapping blocks.  Even windows: the
// unpaired tap's time-domain share is summed the same way in a second array of 2048 floats per slot.  Twelve-wave structure,
// dynamic filter queue: the load balance and occupancy of the parameter-gradient kernel.
constexpr size_t fft_wgg_bwd_dx_lds_bytes(int NW, int K) {
    return fft_wgg_lds_bytes(NW, K) + (size_t)2 * kWgRingFloat2 * 8 + ((K & 1) ? 0 : (size_t)2 * kFftN * 4);
}
template <int NW, int NI, bool HALF = true, bool DX = false>
__global__ __launch_bounds__(NW * 64, (NW + 3) / 4) void leaf_fft_wgg_bwd_kernel(const FftParams p) {
    extern __shared__ __attribute__((aligned(16))) float wsm[];
    float2* twl = reinterpret_cast<float2*>(wsm);
    float2* twh = twl + 32 * 64;
    float2* ring = twh + 64;                                              // A': [2][kWgRingFloat2]
    [[maybe_unused]] float2* gsum = ring + 2 * kWgRingFloat2;            // DX: G, Hermitian-folded: [2][kWgRingFloat2]
    int* q = reinterpret_cast<int*>(ring + (DX ? 4 : 2) * kWgRingFloat2);
    // q: 0 next task | 1,2 spectra stored per slot | 3,4 inverse tasks finished per slot | 5..8 (clip, block) per slot |
    //    9,10 generations released per slot (all readers done) | 11,12 (DX) filters added to the slot's G, ever
    //    | 13,14 (DX, even windows) filters added to the slot's unpaired-tap array, ever
    const int PF = fft_wgg_front_floats(p.K), BP = HALF ? fft_wgg_back_floats(p.K) : fft_wgg_back_floats_full(p.K);
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane0 = tid & 63;
    const bool even = !(p.K & 1);                                         // wave-uniform: the unpaired tap's time-domain terms
    [[maybe_unused]] float* tsum = reinterpret_cast<float*>(q + kWgQueueInts);   // DX, even windows: [2][2048] floats
    float* wbase = reinterpret_cast<float*>(q + kWgQueueInts) + (DX && even ? 2 * kFftN : 0) + (size_t)wave * (PF + kFftN + BP);
    float* scr = wbase + PF;                                              // the row's 2048 samples; transposition scratch in its head
    const unsigned scr_lds = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(__attribute__((address_space(3))) float*)scr);

    fft_build_twiddles_wg(twl, twh, tid, (int)blockDim.x);
    if (tid < kWgQueueInts) q[tid] = 0;
    __syncthreads();

    const int PADL = p.padL, ROT = p.K / 2, LS = p.L, SKr = p.K;
    const int nblocks = p.B * p.nblk;
    const int nset = (nblocks - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x;
    const WgTaskGrid grid = wg_task_grid(p.F, nset);                       // F + 1 slots per set
    const int ntasks = nset > 0 ? 1 + nset * (p.F + 1) : 0;
    auto pull = [&]() {
        int v = 0;
        if (lane0 == 0) v = __hip_atomic_fetch_add(&q[0], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        return __builtin_amdgcn_readfirstlane(v);
    };
    auto decode = [&](int t, int& set, int& role) { wg_task_decode(grid, t, set, role); };
    auto row_of = [&](int role) { return role > 0 && role <= p.F ? role - 1 : 0; };
    float rq[32];
    auto load_real_spectrum = [&](int f, int lane) {
        const float* src = reinterpret_cast<const float*>(p.H) + (size_t)f * kFftN + lane;
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
        if (role == 0 || role > p.F) {
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
                    const int n = n_c - PADL + ((i + ROT) & (kFftN - 1));
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
                    if (even)
                        for (int i = lane; i < kFftN; i += 64) tsum[slot * kFftN + i] = 0.0f;
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
        const int f = role - 1;
        if (set != seen_set) {                                            // this wave's first filter of the block: once the
            wg_wait_ge(&q[1 + slot], gen + 1);                            // spectrum is in the ring it stays until every filter is done
            seen_b = __builtin_amdgcn_readfirstlane(wg_ld(&q[5 + 2 * slot]));
            seen_c = __builtin_amdgcn_readfirstlane(wg_ld(&q[6 + 2 * slot]));
            seen_set = set;
        }
        const int b = seen_b, c = seen_c;
        const int gb = b * p.nblk + c;
        float amu, asg, dpw;
        {
            float dummy_re[32], dummy_im[32];
            wgg_bwd_filter<NI, DX ? 2 : 0, HALF>(p, A, lane, f, b, c, even, rq, wbase, scr, scr_lds, twl, twh, dummy_re, dummy_im, amu,
                                                 asg, dpw, gsum + slot * kWgRingFloat2, &q[11 + slot], gen * p.F + f,
                                                 tsum + slot * kFftN, &q[13 + slot]);
        }
        if constexpr (DX) {
            if (f == p.F - 1) {
                wg_dx_finish<HALF>(p, gsum + slot * kWgRingFloat2, even ? tsum + slot * kFftN : nullptr, gb, ROT, lane, scr, scr_lds, twl, twh);
            }
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
            const float sp = pool_sigma(p.pool_w[f], SKr);
            p.dkpart[((size_t)gb * p.F + f) * 2] = amu;
            p.dkpart[((size_t)gb * p.F + f) * 2 + 1] = asg;
            p.dwpart[(size_t)gb * p.F + f] = dpw / (sp * sp * sp);
        }
        // ---- this task is done with the slot; the wave that finishes the block's last filter releases it
        wg_release();
        int old = 0;
        if (lane == 0) old = __hip_atomic_fetch_add(&q[3 + slot], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        old = __builtin_amdgcn_readfirstlane(old);
        if (old == gen * p.F + p.F - 1) {
            if (lane == 0) __hip_atomic_fetch_add(&q[9 + slot], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
        t = tn;
        set = nset_i;
        role = nrole;
    }
}

}  // namespace
