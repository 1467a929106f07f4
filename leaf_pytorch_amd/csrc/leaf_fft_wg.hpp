// leaf_fft_wg.hpp -- overlap-save forward, second generation: one WORKGROUP per block, spectrum shared through LDS
// Part of the single translation unit leaf_kernels.hip (gfx950 only); see that file's header comment.
//
// Why (round-1 profile of leaf_fft_kernel, profiles/r01): the kernel is fp32-VALU-issue-bound but issues only ~0.49 of
// the SIMD's slots: a wave issues at most one VALU instruction per ~4.6 cycles, so two waves per SIMD cannot fill a
// 2-cycle pipe, and each wave-level FFT has four dependent LDS phases.  Two waves per SIMD was forced by registers: the
// block's spectrum A' (64 VGPRs) stayed resident across the wave's filters next to the 64 registers of the transform.
//
// Here A' lives in LDS instead, computed ONCE per block and read by every wave of the workgroup at the spectral
// multiply, which (i) frees 64 + 32 VGPRs -> three waves per SIMD (12-wave workgroups, <= 168 VGPRs), (ii) removes the
// repeated forward transforms (one per block instead of one per (block, filter group)), and (iii) needs only half the
// spectrum: the (rotated) block is real, so A'[N - k] = conj(A'[k]) and bins 0..1024 are stored (8.2 KB per block);
// the upper half is read back mirrored (a descending ds_read_b64, conflict-free like the ascending one).
//
// Scheduling: no barriers.  A workgroup walks its blocks (blockIdx.x, + gridDim.x, ...) through a task queue in LDS:
//     fwd(0), [fwd(1), inv(0,0) .. inv(0,F-1)], [fwd(2), inv(1,0) ..], ...
// pulled in order with one LDS atomic per task.  fwd(i) = load + forward transform of set i's block into ring slot i & 1;
// inv(i,f) = spectral multiply with filter f + inverse transform + |.|^2 + pooling (the arithmetic of
// leaf_fft_kernel's static-geometry path; results differ from it only by the rounding of the mirrored upper half-spectrum,
// ~1e-7 relative, so a clip is bit-identical across batches served by THIS kernel, not across the two kernels).  Dependencies are two monotonic counters per slot:
//     inv(i,f) waits for  fwd_cnt[slot] >= (i >> 1) + 1          (the spectrum is there)
//     fwd(i)   waits for  inv_cnt[slot] >= (i >> 1) * F          (every reader of the slot's previous occupant is done)
// Because fwd(i+1) is queued BEFORE set i's inverse tasks, one wave computes the next spectrum while the other eleven
// work on the current block, and nobody waits for it.  Deadlock-free: a task only ever waits for tasks queued before it.
#pragma once
#include "leaf_fft.hpp"

namespace {

constexpr int kWgRingFloat2 = 1032;            // bins 0..1024 of a block's spectrum, padded
constexpr int kWgQueueInts = 16;               // q_next, fwd_cnt[2], inv_cnt[2]

__device__ __forceinline__ int wg_ld(const int* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
__device__ __forceinline__ void wg_wait_ge(const int* p, int need) {
    while (wg_ld(p) < need) __builtin_amdgcn_s_sleep(1);
    asm volatile("" ::: "memory");
}

// floats of dynamic LDS for NW waves and a static pooling row of GU floats
constexpr int fft_wg_row_floats(int SK) { return (kGPad + SK + 63 + 3) / 4 * 4; }
constexpr size_t fft_wg_lds_bytes(int NW, int SK) {
    return ((size_t)kTwFloats + 2 * 2 * kWgRingFloat2 + kWgQueueInts + (size_t)NW * (32 * 65 + fft_wg_row_floats(SK))) * 4;
}

template <int SK, int SHOP, int NW>
__global__ __launch_bounds__(NW * 64, 3) void leaf_fft_wg_kernel(const FftParams p) {
    extern __shared__ __attribute__((aligned(16))) float wsm[];
    float2* twl = reinterpret_cast<float2*>(wsm);                        // [32][64]
    float2* twh = twl + 32 * 64;                                          // [32][2]
    float2* ring = twh + 64;                                              // [2][kWgRingFloat2]
    int* q = reinterpret_cast<int*>(ring + 2 * kWgRingFloat2);            // q_next | fwd_cnt[2] | inv_cnt[2]
    constexpr int GU = fft_wg_row_floats(SK);
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane0 = tid & 63;
    float* scr = reinterpret_cast<float*>(q + kWgQueueInts) + (size_t)wave * (32 * 65 + GU);
    float* sG = scr + 32 * 65;

    fft_build_twiddles(twl, twh, tid, NW * 64);
    if (tid < kWgQueueInts) q[tid] = 0;
    __syncthreads();

    constexpr int PADL = SK / 2 + SK % 2 - 1;
    constexpr int LS = fft_block_len(SK, SHOP, true);
    constexpr int DMIN = -((SK - 1 - PADL) / SHOP);
    constexpr int DMAX = (LS - 1 + PADL) / SHOP;
    constexpr int NFR = DMAX - DMIN + 1;
    constexpr int NROW = LS / 64;
    constexpr int NGRP = (NFR + 15) / 16;
    static_assert(LS % SHOP == 0 && LS % 64 == 0 && LS > 0 && NFR <= 32 && (SK & 1), "static odd-window geometry");

    const int nblocks = p.B * p.nblk;
    const int nset = (nblocks - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x;      // blocks of this workgroup
    const int ntasks = nset > 0 ? 1 + nset * (p.F + 1) : 0;
    auto pull = [&]() {
        int v = 0;
        if (lane0 == 0) v = __hip_atomic_fetch_add(&q[0], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        return __builtin_amdgcn_readfirstlane(v);
    };
    // task -> (set, role): role 0 = forward transform of set `set`, role r >= 1 = filter r - 1 of set `set`
    auto decode = [&](int t, int& set, int& role) {
        if (t == 0) { set = 0; role = 0; return; }
        const int u = t - 1;
        set = u / (p.F + 1);
        role = u - set * (p.F + 1);
        if (role == 0) set += 1;                                          // the NEXT set's spectrum, ahead of this set's filters
    };
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
    int t = pull(), set = 0, role = 0;
    if (t < ntasks) decode(t, set, role);
    load_real_spectrum(role > 0 ? role - 1 : 0, lane0);
    while (t < ntasks) {
        int lane = lane0;
        asm volatile("" : "+v"(lane));
        const int slot = set & 1, gen = set >> 1;
        float2* A = ring + slot * kWgRingFloat2;
        const int gb = (int)blockIdx.x + set * (int)gridDim.x;
        const int b = gb / p.nblk, c = gb - b * p.nblk;
        const int n_c = c * p.L;
        if (role == 0) {
            // ---- forward transform of block gb into ring slot `slot` (skipped past the last set)
            if (set < nset) {
                float are[32], aim[32];
                const float* xb = static_cast<const float*>(p.x) + (size_t)b * p.T;
                const unsigned short* xh = static_cast<const unsigned short*>(p.x) + (size_t)b * p.T;
                if (p.io_bf16) {
#pragma unroll
                    for (int r = 0; r < 32; ++r) {
                        const int i = 64 * r + lane;                      // block rotated left by padL samples
                        const int n = n_c - p.padL + ((i + p.padL) & (kFftN - 1));
                        const unsigned v = xh[min(max(n, 0), p.T - 1)];
                        are[r] = (n >= 0 && n < p.T) ? __uint_as_float(v << 16) : 0.0f;
                        aim[r] = 0.0f;
                    }
                } else {
#pragma unroll
                    for (int r = 0; r < 32; ++r) {
                        const int i = 64 * r + lane;
                        const int n = n_c - p.padL + ((i + p.padL) & (kFftN - 1));
                        are[r] = (n >= 0 && n < p.T) ? xb[n] : 0.0f;
                        aim[r] = 0.0f;
                    }
                }
                fft2048(are, aim, scr, twl, twh, lane);                   // register i <-> bin 64 brev5(i) + lane
                wg_wait_ge(&q[3 + slot], gen * p.F);                      // the slot's previous readers are done
#pragma unroll
                for (int i = 0; i < 32; ++i) {
                    const int k = brev5(i);
                    if (k < 16) A[64 * k + lane] = make_float2(are[i], aim[i]);
                    else if (k == 16 && lane == 0) A[1024] = make_float2(are[i], aim[i]);
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                if (lane == 0) __hip_atomic_fetch_add(&q[1 + slot], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
            // rq is redefined UNCONDITIONALLY here (row 0 when the next task is not an inverse one), so that the previous
            // row is dead throughout this branch -- carried through the forward transform it would be spilled every task
            t = pull();
            if (t < ntasks) decode(t, set, role);
            else role = 0;
            load_real_spectrum(role > 0 ? role - 1 : 0, lane);
            continue;
        }
        // ---- filter f of block gb
        const int f = role - 1;
        const int Lv = min(p.L, p.T - n_c);
        int mlo = n_c + p.padL - p.K + 1;                                 // first frame whose window reaches the block
        mlo = mlo <= 0 ? 0 : (mlo + p.hop - 1) / p.hop;
        const int mhi = min(p.TP - 1, (n_c + Lv - 1 + p.padL) / p.hop);
        wg_wait_ge(&q[1 + slot], gen + 1);                                // the block's spectrum is in the ring
        // Z = conj(A' R_f): rows 0..15 straight from the ring, rows 16..31 mirrored (A'[N - e] = conj(A'[e]))
        // (8-row chunks, fenced: all 32 ring reads in flight at once would need 64 registers next to rq and Z)
        float zre[32], zim[32];
#pragma unroll
        for (int k0 = 0; k0 < 32; k0 += 8) {
            float2 a[8];
            asm volatile("" ::: "memory");
#pragma unroll
            for (int j = 0; j < 8; ++j) a[j] = k0 < 16 ? A[64 * (k0 + j) + lane] : A[kFftN - 64 * (k0 + j) - lane];
            asm volatile("" ::: "memory");
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int k = k0 + j;
                zre[k] = a[j].x * rq[k];
                zim[k] = k0 < 16 ? -(a[j].y * rq[k]) : a[j].y * rq[k];
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::"v"(zre[31]), "v"(zim[31]) : "memory");
        if (lane == 0) __hip_atomic_fetch_add(&q[3 + slot], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        // pooling row of this filter -> wave-private LDS (16 bytes per lane per instruction), lands under the transform
        {
            const float* gsrc = p.Gz + (size_t)f * p.GZ;
#pragma unroll
            for (int i0 = 0; i0 < GU; i0 += 256)
                if (i0 + 256 <= GU || i0 + 4 * lane < GU)
                    __builtin_amdgcn_global_load_lds(gsrc + i0 + 4 * lane, (__attribute__((address_space(3))) void*)(sG + i0), 16, 0, 0);
            asm volatile("" ::: "memory");
        }
        fft2048(zre, zim, scr, twl, twh, lane);                           // register i <-> samples 64 brev5(i) + lane
        float er[NROW];
#pragma unroll
        for (int i = 0; i < 32; ++i) {
            const int r = brev5(i);
            if (r < NROW) er[r] = 64 * r + lane < Lv ? zre[i] * zre[i] + zim[i] * zim[i] : 0.0f;
        }
        // next task: reserved now so that its filter's spectrum row streams in under the pooling
        const int tn = pull();
        int nset_i = 0, nrole = 0;
        if (tn < ntasks) decode(tn, nset_i, nrole);
        asm volatile("" ::"v"(er[0]), "v"(er[NROW - 1]));
        load_real_spectrum(nrole > 0 ? nrole - 1 : 0, lane);             // (row 0 as a dummy when there is no next filter)
        asm volatile("s_waitcnt vmcnt(32)" ::: "memory");                 // the row DMA (issued before the 32 loads) has landed
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
                    acc[fi / 16][fi % 16] = fmaf(er[r], sG[kGPad + 64 * r - is + lane], acc[fi / 16][fi % 16]);
            }
        }
        asm volatile("" : "+v"(acc[0][0]));
#pragma unroll
        for (int g = 0; g < NGRP; ++g) {
            const float v = frame_butterfly16(acc[g], lane);
            const int fi = 16 * g + ((lane >> 5) & 1) * 8 + ((lane >> 4) & 1) * 4 + ((lane >> 3) & 1) * 2 + ((lane >> 2) & 1);
            const int m = n_c / SHOP + DMIN + fi;
            if ((lane & 3) == 0 && fi < NFR && m >= mlo && m <= mhi) {
                const int first_block = max(0, m * p.hop - p.padL) / p.L;
                p.part[(((size_t)b * p.F + f) * p.nslot + (c - first_block)) * p.TP + m] = v;
            }
        }
        // the pooling's LDS reads of sG must be complete before the next task's row DMA overwrites the buffer
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        t = tn;
        set = nset_i;
        role = nrole;
    }
}

}  // namespace
