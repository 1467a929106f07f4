// leaf_fft_wgg.hpp -- the workgroup-per-block overlap-save forward (leaf_fft_wg.hpp) for ANY window the 2048-sample plan
// covers: window, hop and block length at run time, odd and even windows.
// Part of the single translation unit leaf_kernels.hip (gfx950 only); see that file's header comment.
//
// Same task queue, spectrum ring and LDS-lean transform as leaf_fft_wg_kernel; what differs is the pooling, which cannot
// unroll over compile-time frame offsets.  Round 1's generic pooling (leaf_fft.hpp: per frame, a switch on the window's
// first row into unrolled row code) costs +60 % of the kernel at the default geometry.  Here the roles are swapped: the
// block's |y|^2 go to a wave-private LDS row (32 ds_write_b32, natural order, zeros where the block has no sample, K - 1
// zeros in front) and the WEIGHTS stay in registers -- lane l holds taps l, 64 + l, ... (NI = ceil(K / 64) registers, read
// once per task from the filter's table row).  A frame is then NI ds_read_b32 at immediate offsets from one per-frame
// address (window start + lane) and NI FMAs; four frames share one wave reduction (wave_sum4_rows) and four lanes store
// them.  No per-row bookkeeping, no compile-time geometry; per task this is the same number of LDS reads and FMAs as the
// static kernels' unrolled pooling.  The transposition scratch of the transform aliases the head of the energy row.
// Even windows: the K - 1 taps t = -(K/2 - 1) .. K/2 - 1 are Hermitian about t = 0 and go through a real spectrum (block
// rotated by K/2); the unpaired tap t = -K/2 is a scaled copy of the input added after the inverse transform
// (frontend.py:38 gives even windows at 22.05 / 11.025 kHz: K = 552 / 276).
#pragma once
#include "leaf_fft_wg.hpp"

namespace {

// weight registers of the instantiation that serves window K (buckets: one kernel per bucket and parity)
constexpr int fft_wgg_taps_per_lane(int K) {
    return K <= 320 ? 5 : K <= 448 ? 7 : K <= 576 ? 9 : K <= 640 ? 10 : K <= 832 ? 13 : K <= 1024 ? 16 : 19;
}
// wave-private LDS floats: [K - 1 zeros][2048 energies, the first 16 x 68 double as the transposition scratch][zeros the
// last frame's reads run into: taps 64 NI - 1 >= K - 1]
constexpr int fft_wgg_front_floats(int K) { return (K - 1 + 3) / 4 * 4; }
constexpr int fft_wgg_back_floats(int K) { return (64 * fft_wgg_taps_per_lane(K) - K + 3) / 4 * 4 + 4; }
constexpr size_t fft_wgg_wave_floats(int K) { return (size_t)fft_wgg_front_floats(K) + kFftN + fft_wgg_back_floats(K); }
// with the FULL transposition scratch (32 x 68 floats, aliasing the head of the row like the half-size one) the row's data
// and back padding must hold 2176 floats
constexpr int fft_wgg_back_floats_full(int K) { return fft_wgg_back_floats(K) > kWgScrFloats - kFftN ? fft_wgg_back_floats(K) : kWgScrFloats - kFftN; }
constexpr size_t fft_wgg_lds_bytes_full(int NW, int K) {
    return ((size_t)kTwFloats + 2 * 2 * kWgRingFloat2 + kWgQueueInts +
            (size_t)NW * ((size_t)fft_wgg_front_floats(K) + kFftN + fft_wgg_back_floats_full(K))) * 4;
}
constexpr size_t fft_wgg_lds_bytes(int NW, int K) {
    return ((size_t)kTwFloats + 2 * 2 * kWgRingFloat2 + kWgQueueInts + (size_t)NW * fft_wgg_wave_floats(K)) * 4;
}

template <int NW, int NI, bool HALF = true>
__global__ __launch_bounds__(NW * 64, (NW + 3) / 4) void leaf_fft_wgg_kernel(const FftParams p) {
    // HALF: half-size (column form) transposition scratch; else the full one, where the LDS holds it (host's choice)
    extern __shared__ __attribute__((aligned(16))) float wsm[];
    float2* twl = reinterpret_cast<float2*>(wsm);                        // [32][64]
    float2* twh = twl + 32 * 64;                                          // [32][2]
    float2* ring = twh + 64;                                              // [2][kWgRingFloat2]
    int* q = reinterpret_cast<int*>(ring + 2 * kWgRingFloat2);            // q_next | fwd_cnt[2] | inv_cnt[2]
    const int PF = fft_wgg_front_floats(p.K), BP = HALF ? fft_wgg_back_floats(p.K) : fft_wgg_back_floats_full(p.K);
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane0 = tid & 63;
    float* wbase = reinterpret_cast<float*>(q + kWgQueueInts) + (size_t)wave * (PF + kFftN + BP);
    float* scr = wbase + PF;                                              // energies [0, 2048); transposition scratch in its head
    const unsigned scr_lds = __builtin_amdgcn_readfirstlane(
        (unsigned)(size_t)(__attribute__((address_space(3))) float*)scr);   // LDS byte address of this wave's scr

    fft_build_twiddles_wg(twl, twh, tid, (int)blockDim.x);                   // the host may launch fewer than NW waves (LDS)
    if (tid < kWgQueueInts) q[tid] = 0;
    for (int i = lane0; i < PF; i += 64) wbase[i] = 0.0f;                 // written once: nothing else touches the paddings
    for (int i = lane0; i < BP; i += 64) scr[kFftN + i] = 0.0f;
    __syncthreads();

    // geometry at run time: any window the 2048-sample plan covers (even windows: Hermitian K - 1 taps
    // through the real spectrum + the unpaired tap t = -K/2 in the time domain)
    const int PADL = p.padL, ROT = p.K / 2, LS = p.L, SKr = p.K, SHOPr = p.hop;
    const bool even = !(p.K & 1);                                         // wave-uniform: the unpaired tap's time-domain term

    // Task ids: F + 1 slots per set (wg_task_decode); slot 0 of set i is fwd(i + 1), slots 1..F are the set's filters.
    // blocks dealt contiguously; clips all of whose blocks this workgroup ran are finalized in its tail (as leaf_fft_wg_kernel)
    const OwnedClips deal{p.B * p.nblk, (int)gridDim.x, p.nblk};
    const int first_gb = deal.start((int)blockIdx.x);
    const int nset = deal.count((int)blockIdx.x);      // blocks of this workgroup
    const WgTaskGrid grid = wg_task_grid(p.F, nset);                       // F + 1 slots per set
    const int ntasks = nset > 0 ? 1 + nset * (p.F + 1) : 0;
    auto pull = [&]() {
        int v = 0;
        if (lane0 == 0) v = __hip_atomic_fetch_add(&q[0], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        return __builtin_amdgcn_readfirstlane(v);
    };
    // task -> (set, role): role 0 = forward transform of set `set`, role 1..F = filter role - 1 of set `set`
    auto decode = [&](int t, int& set, int& role) { wg_task_decode(grid, t, set, role); };
    auto row_of = [&](int role) { return role > 0 && role <= p.F ? role - 1 : 0; };   // spectrum row to prefetch
    float rq[32];                                                         // R_f[64 k + lane], natural row order
    auto load_real_spectrum = [&](int f, int lane) {
        const float* src = reinterpret_cast<const float*>(p.H) + (size_t)f * kFftN + lane;
        asm volatile("" ::: "memory");
#pragma unroll
        for (int k = 0; k < 32; ++k) rq[k] = src[64 * k];
        asm volatile("" ::: "memory");
    };

    // Invariant at the loop head: (set, role) is the decoded current task, and when it is an inverse task its filter's
    // spectrum row has already been requested into rq (by the previous task, under its pooling).
    int seen_set = -1, seen_b = 0, seen_c = 0;                            // block coordinates of the set this wave last worked on
    int t = pull(), set = 0, role = 0;
    if (t < ntasks) decode(t, set, role);
    load_real_spectrum(row_of(role), lane0);
    while (t < ntasks) {
        int lane = lane0;
        asm volatile("" : "+v"(lane));
        const int slot = set & 1, gen = set >> 1;
        float2* A = ring + slot * kWgRingFloat2;
        if (role == 0) {
            // ---- forward transform of block gb into ring slot `slot` (skipped past the last set)
            if (set < nset) {
                const int gb = first_gb + set;
                const int b = gb / p.nblk, c = gb - b * p.nblk;               // the only division per block
                const int n_c = c * LS;
                float are[32], aim[32];
                const float* xb = static_cast<const float*>(p.x) + (size_t)b * p.T;
                const unsigned short* xh = static_cast<const unsigned short*>(p.x) + (size_t)b * p.T;
                if (p.io_bf16) {
#pragma unroll
                    for (int r = 0; r < 32; ++r) {
                        const int i = 64 * r + lane;                      // block rotated left by padL samples
                        const int n = n_c - PADL + ((i + ROT) & (kFftN - 1));
                        const unsigned v = xh[min(max(n, 0), p.T - 1)];
                        are[r] = (n >= 0 && n < p.T) ? __uint_as_float(v << 16) : 0.0f;
                        aim[r] = 0.0f;
                    }
                } else {
#pragma unroll
                    for (int r = 0; r < 32; ++r) {
                        const int i = 64 * r + lane;
                        const int n = n_c - PADL + ((i + ROT) & (kFftN - 1));
                        are[r] = (n >= 0 && n < p.T) ? xb[n] : 0.0f;
                        aim[r] = 0.0f;
                    }
                }
                fft2048w<HALF>(are, aim, scr, scr_lds, twl, twh, lane);        // register i <-> bin 64 brev5(i) + lane
                wg_wait_ge(&q[3 + slot], gen * p.F);                      // the slot's previous readers are done
#pragma unroll
                for (int i = 0; i < 32; ++i) {
                    const int k = brev5(i);
                    if (k < 16) A[64 * k + lane] = make_float2(are[i], aim[i]);
                    else if (k == 16 && lane == 0) A[1024] = make_float2(are[i], aim[i]);
                }
                if (lane == 0) { q[5 + 2 * slot] = b; q[6 + 2 * slot] = c; }      // the block's coordinates, for its readers
                wg_release();
                if (lane == 0) __hip_atomic_fetch_add(&q[1 + slot], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
            // rq is redefined UNCONDITIONALLY here (row 0 when the next task is not an inverse one), so that the previous
            // row is dead throughout this branch -- carried through the forward transform it would be spilled every task
            t = pull();
            if (t < ntasks) decode(t, set, role);
            else role = 0;
            load_real_spectrum(row_of(role), lane);
            continue;
        }
        // ---- filter f of the block in ring slot `slot`
        const int f = role - 1;
        if (set != seen_set) {                                            // this wave's first filter of the block: once the
            wg_wait_ge(&q[1 + slot], gen + 1);                            // spectrum is in the ring it stays until every filter is done
            seen_b = __builtin_amdgcn_readfirstlane(wg_ld(&q[5 + 2 * slot]));
            seen_c = __builtin_amdgcn_readfirstlane(wg_ld(&q[6 + 2 * slot]));
            seen_set = set;
        }
        const int b = seen_b, c = seen_c;
        const int n_c = c * LS;
        const int Lv = min(LS, p.T - n_c);
        int mlo = n_c + PADL - SKr + 1;                                   // first frame whose window reaches the block
        mlo = mlo <= 0 ? 0 : (mlo + SHOPr - 1) / SHOPr;
        const int mhi = min(p.TP - 1, (n_c + Lv - 1 + PADL) / SHOPr);
        // Z = conj(A' R_f): rows 0..15 straight from the ring, rows 16..31 mirrored (A'[N - e] = conj(A'[e]))
        // (8-row chunks, fenced: all 32 ring reads in flight at once would need 64 registers next to rq and Z)
        float zre[32], zim[32];
        {
            // two streams of 16 rows: ascending from A[lane], and the mirror A[2048 - 64 k - lane], k = 16..31, read as
            // rows 15..0 of the base A[1088 - lane] (= k = 31 first); lds_stream32 walks 2 x 16 rows
            const unsigned a_lo = lds_addr(A + lane), a_hi = lds_addr(A + (kFftN - 64 * 31) - lane);
            v2f lo[16], hi[16];
            auto rd = [&](auto kk) {
                constexpr int k = decltype(kk)::value;
                if constexpr (k < 16) lds_rd8<512 * k>(lo[k], a_lo);
                else lds_rd8<512 * (31 - k)>(hi[k - 16], a_hi);
            };
            (void)rd;
            // chunk 0: rows 0..7, chunk 1: rows 8..15, chunk 2: rows 16..23, chunk 3: rows 24..31
#define LEAF_RD8(B) rd(std::integral_constant<int, B + 0>{}); rd(std::integral_constant<int, B + 1>{}); \
                    rd(std::integral_constant<int, B + 2>{}); rd(std::integral_constant<int, B + 3>{}); \
                    rd(std::integral_constant<int, B + 4>{}); rd(std::integral_constant<int, B + 5>{}); \
                    rd(std::integral_constant<int, B + 6>{}); rd(std::integral_constant<int, B + 7>{});
            v2f(&lo0)[8] = *reinterpret_cast<v2f(*)[8]>(&lo[0]);
            v2f(&lo1)[8] = *reinterpret_cast<v2f(*)[8]>(&lo[8]);
            v2f(&hi0)[8] = *reinterpret_cast<v2f(*)[8]>(&hi[0]);
            v2f(&hi1)[8] = *reinterpret_cast<v2f(*)[8]>(&hi[8]);
#if LEAF_FFT32_DIT && LEAF_FFT_FUSE_TWIDDLE
            // the spectral multiply fused with the first decimation-in-time stage of the transform (pairs of rows (k, k + 16),
            // unit twiddles): with za = conj(A'[k]) R[k] and zb = the mirrored row's product,
            //     out[k] = za + zb,  out[k + 16] = za - zb   as one product and two FMAs per component -- 6 instructions per pair
            // instead of 4 products + 4 additions.  Mirrored rows are read as rows 15..0 of a_hi: row k + 16 is hi[k].
            auto pair = [&](int k) {
                const float ra = rq[k], rb = rq[k + 16];
                const float tr_ = lo[k].x * ra, ti_ = -(lo[k].y * ra);
                zre[k] = fmaf(hi[k].x, rb, tr_);
                zim[k] = fmaf(hi[k].y, rb, ti_);
                zre[k + 16] = fmaf(-hi[k].x, rb, tr_);
                zim[k + 16] = fmaf(-hi[k].y, rb, ti_);
            };
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
#else
            LEAF_RD8(0) LEAF_RD8(8)
            lds_wait8<8>(lo0);
#pragma unroll
            for (int k = 0; k < 8; ++k) { zre[k] = lo[k].x * rq[k]; zim[k] = -(lo[k].y * rq[k]); }
            LEAF_RD8(16)
            lds_wait8<8>(lo1);
#pragma unroll
            for (int k = 8; k < 16; ++k) { zre[k] = lo[k].x * rq[k]; zim[k] = -(lo[k].y * rq[k]); }
            LEAF_RD8(24)
            lds_wait8<8>(hi0);
#pragma unroll
            for (int k = 16; k < 24; ++k) { zre[k] = hi[k - 16].x * rq[k]; zim[k] = hi[k - 16].y * rq[k]; }
            lds_wait8<0>(hi1);
#pragma unroll
            for (int k = 24; k < 32; ++k) { zre[k] = hi[k - 16].x * rq[k]; zim[k] = hi[k - 16].y * rq[k]; }
#endif
#undef LEAF_RD8
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::"v"(zre[31]), "v"(zim[31]) : "memory");
        wg_release();
        if (lane == 0) __hip_atomic_fetch_add(&q[3 + slot], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        fft2048w<HALF, LEAF_FFT32_DIT && LEAF_FFT_FUSE_TWIDDLE>(zre, zim, scr, scr_lds, twl, twh, lane);   // register i <-> samples 64 brev5(i) + lane
        if (even) {
            // the unpaired tap t = -K/2: y[cL + r] += w[-K/2] x[cL - padL + r]; with u = conj(y) in registers (register i <->
            // sample 64 brev5(i) + lane): u += conj(c) a[r].  The block samples come back from L2 in 8-row chunks.
            const float cre = p.lone[2 * f], cim = p.lone[2 * f + 1];
            const int nb_x = n_c - PADL;
            const bool x_interior = nb_x >= 0 && nb_x + kFftN <= p.T;
            pin32(zre);
            pin32(zim);
#pragma unroll
            for (int i0 = 0; i0 < 32; i0 += 8) {
                float xa[8];
                block_rows8(p.x, (size_t)b * p.T, p.io_bf16, nb_x, p.T, lane, x_interior, [&](int j) { return brev5(i0 + j); }, xa);
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
        // the filter's pooling taps: lane l holds g_f[l], g_f[64 + l], ... (the table row is zero past tap K - 1)
        pin32(zre);
        pin32(zim);
        float w[NI];
        {
            const float* gsrc = p.Gz + (size_t)f * p.GZ + kGPad;
            int ofs = 0;
            asm volatile("" : "+v"(ofs) : : "memory");                    // opaque: keeps the loads below the transform
#pragma unroll
            for (int i = 0; i < NI; ++i) w[i] = gsrc[min(64 * i + lane, p.GZ - kGPad - 1) + ofs];
        }
        // |y|^2 -> the wave's energy row, natural order, zero where the block has no sample (rows past the block length are
        // written too: the transform's scratch aliases the head of the row, and late windows read its tail)
        {
            using lds_fp = __attribute__((address_space(3))) float*;
            const lds_fp erow = (lds_fp)scr + lane;
            const int nrow = LS >> 6;                                     // >= 13: K <= 1216
            if (Lv == LS) {                                               // all but a clip's last block: whole rows, no lane masks
#pragma unroll
                for (int i = 0; i < 32; ++i) {
                    const int r = brev5(i);
                    const float e = zre[i] * zre[i] + zim[i] * zim[i];
                    erow[64 * r] = (r < 13 || r < nrow) ? e : 0.0f;
                }
            } else {
#pragma unroll
                for (int i = 0; i < 32; ++i) {
                    const int r = brev5(i);
                    erow[64 * r] = 64 * r + lane < Lv ? zre[i] * zre[i] + zim[i] * zim[i] : 0.0f;
                }
            }
        }
        // next task: reserved now so that its filter's spectrum row streams in under the pooling
        const int tn = pull();
        int nset_i = 0, nrole = 0;
        if (tn < ntasks) decode(tn, nset_i, nrole);
        load_real_spectrum(row_of(nrole), lane);
        asm volatile("s_waitcnt vmcnt(32)" ::: "memory");                 // the taps (issued before the 32 loads) have landed
        // ---- Gaussian pooling: frame m = sum_j g_f[j] e[is_m + j], is_m = m hop - padL - n_c (window start relative to the
        // block; the zeros in front of and behind the energies stand for the parts of the window outside it).  Four frames
        // per turn: 4 NI reads in flight, 4 NI FMAs, one reduction; frames past mhi repeat frame mhi and are not stored.
        {
            using lds_cfp = const __attribute__((address_space(3))) float*;
            const lds_cfp ebase = (lds_cfp)scr + lane;
            const int is0 = -PADL - n_c;                                  // window start of frame m relative to the block: m hop + is0
#pragma nounroll
            for (int m4 = mlo; m4 <= mhi; m4 += 4) {
                float a[4];
                lds_cfp pk[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    pk[k] = ebase + (min(m4 + k, mhi) * SHOPr + is0);
                    a[k] = 0.0f;
                }
#pragma unroll
                for (int i = 0; i < NI; ++i)
#pragma unroll
                    for (int k = 0; k < 4; ++k) a[k] = fmaf(w[i], pk[k][64 * i], a[k]);
                const float v = wave_sum4_rows(a[0], a[1], a[2], a[3]);   // 16-lane row q holds frame m4 + q
                const int m = m4 + (lane >> 4);
                if ((lane & 15) == 0 && m <= mhi) {
                    // slot = blocks between the one holding the window's first sample and this one (no division: <= 3 steps)
                    const int first = max(0, m * SHOPr - PADL);
                    int back = 0;
                    for (int nb = n_c; first < nb; nb -= LS) ++back;
                    p.part[(((size_t)b * p.F + f) * p.nslot + back) * p.TP + m] = v;
                }
            }
        }
        // the pooling's LDS reads must be complete before the next task's transform reuses the head of the row as scratch
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        t = tn;
        set = nset_i;
        role = nrole;
    }
    if (p.fin_fused)                                                      // the waves' rows are free: tile memory of the tail
        wg_tail_finalize<32>(p.fin, (first_gb + p.nblk - 1) / p.nblk, (first_gb + nset) / p.nblk,
                             reinterpret_cast<float*>(q + kWgQueueInts), tid, (int)blockDim.x);
}

}  // namespace
