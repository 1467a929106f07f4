// leaf_fft.hpp -- FFT (overlap-save) formulation of the fused forward: one wave = one 2048-sample block
// Part of the single translation unit leaf_kernels.hip (gfx950 only); see that file's header comment.
//
// Why: the filterbank is a length-K cross-correlation per filter.  In direct (MFMA) form it costs ~2*K flops per
// filter-sample even after the Hermitian halving; by the convolution theorem a block of N = 2048 input samples costs
// one forward FFT (shared by all filters) plus, per filter, a spectral multiply and one inverse FFT -- ~70 flops per
// filter-sample at K = 401, about 10x fewer.  Everything stays fp32; parity with the reference is the same 2e-6 class
// (tests/test_gpu_parity.py runs the goldens through this path too).
//
// Block layout (overlap-save): block c of clip b produces outputs n in [cL, cL+L), L = fft_block_len(K, hop), from
// a[i] = xz[cL - padL + i], i < N:  y[cL + r] = sum_j w[j] a[r+j] = IFFT(FFT(a) * Hr)[r],  Hr[k] = sum_j w[j] e^{+2 pi i jk/N}
// (w = the taps exactly as convolution.py:88-90 hands them to conv1d; no tap cut is needed here).  For odd K the taps
// are Hermitian about the centre tap, so in zero-phase layout their spectrum is real and the block is loaded rotated
// instead (see leaf_fft_kernel).
//
// Wave-level FFT: lane l, register r hold element 64 r + l.  Four-step decomposition N = 32 x 64:
//   32-point FFT over r in registers -> twiddle W_N^{l k1} -> transpose through wave-private LDS -> radix-2 across
//   the two half-waves (v_permlane32_swap) -> 32-point FFT in registers.  Output element 64 k' + l sits on lane l
//   again (register brev5-permuted, a compile-time relabel), so the inverse transform (conjugate trick) needs no
//   re-layout.  Radix-2 butterflies with compile-time twiddles (decimation in time, FMA-fused); the N-point twiddles
//   come from an LDS table.
#pragma once
#include "leaf_common.hpp"
#include "leaf_fused.hpp"      // SlotGeom, the PCEN row helpers

#ifndef LEAF_FFT32_DIT
#define LEAF_FFT32_DIT 1               // 32-point register transforms: 1 decimation in time with FMA-fused butterflies, 0 DIF
#endif
#ifndef LEAF_FFT_SWAP
#define LEAF_FFT_SWAP 1                // half-wave exchange of the wave-level FFT: 1 v_permlane32_swap (VALU), 0 ds_bpermute
#endif
#ifndef LEAF_FFT_ABLATE
#define LEAF_FFT_ABLATE 0              // measurement only (tools/ablate_fft.py; results are wrong by construction):
#endif                                 // bit 0 no spectrum loads, 1 no inverse transform, 2 no pooling FMAs / LDS reads,
                                       // 3 serial instead of butterfly reduction, 4 no pooling-row DMA, 5 every filter
                                       // reads spectrum row 0 (L1-resident), 6 no partial-sum store
namespace {

constexpr int kFftN = 2048;
constexpr int kFftWaves = 8;             // waves per workgroup (2 per SIMD); every wave is an independent worker
constexpr int kGPad = 64;                // zero padding in front of each pooling-window row
constexpr int kFftFQ = 10;               // most filters per task (one forward transform serves them all)

__host__ __device__ constexpr int brev5(int i) {
    return ((i & 1) << 4) | ((i & 2) << 2) | (i & 4) | ((i & 8) >> 2) | ((i & 16) >> 4);
}

template <int HALF>
__device__ __forceinline__ void fft32_stage(float (&re)[32], float (&im)[32]) {
    constexpr float C[16] = {1.0f, 0.98078528f, 0.923879533f, 0.831469612f, 0.707106781f, 0.555570233f, 0.382683432f, 0.195090322f, 0.0f, -0.195090322f, -0.382683432f, -0.555570233f, -0.707106781f, -0.831469612f, -0.923879533f, -0.98078528f};
    constexpr float S[16] = {0.0f, -0.195090322f, -0.382683432f, -0.555570233f, -0.707106781f, -0.831469612f, -0.923879533f, -0.98078528f, -1.0f, -0.98078528f, -0.923879533f, -0.831469612f, -0.707106781f, -0.555570233f, -0.382683432f, -0.195090322f};
#pragma unroll
    for (int blk = 0; blk < 32; blk += 2 * HALF) {
#pragma unroll
        for (int j = 0; j < HALF; ++j) {
            const int a = blk + j, b = a + HALF;
            constexpr int STEP = 16 / HALF;
            const int tw = j * STEP;
            const float ur = re[a] + re[b], ui = im[a] + im[b];
            const float vr = re[a] - re[b], vi = im[a] - im[b];
            re[a] = ur;
            im[a] = ui;
            if (tw == 0) {
                re[b] = vr;
                im[b] = vi;
            } else if (tw == 8) {                        // W = -i
                re[b] = vi;
                im[b] = -vr;
            } else {
                re[b] = vr * C[tw] - vi * S[tw];
                im[b] = vr * S[tw] + vi * C[tw];
            }
        }
    }
}

// 32-point forward DFT (e^{-2 pi i nk/32}) in registers: register i ends up holding X[brev5(i)].
//
// LEAF_FFT32_DIT = 1 (default): radix-2 decimation in TIME on the bit-reversed register labelling (position p of the
// flow graph lives in register brev5(p) -- a compile-time relabel -- so input and output conventions are those of the
// DIF version).  A DIT butterfly is a +- w b, which fuses with the twiddle product: a + w b takes two FMAs per
// component, and a - w b = 2a - (a + w b) one more: 6 instructions instead of the DIF butterfly's 8 (subtract, then a
// 4-instruction complex multiply).  388 instead of 456 instructions per transform; the kernel is VALU-issue-bound, so
// this is time.  LEAF_FFT32_DIT = 0: the decimation-in-frequency version.
template <int HALF>
__device__ __forceinline__ void fft32_dit_stage(float (&re)[32], float (&im)[32]) {
    constexpr float C[16] = {1.0f, 0.98078528f, 0.923879533f, 0.831469612f, 0.707106781f, 0.555570233f, 0.382683432f, 0.195090322f, 0.0f, -0.195090322f, -0.382683432f, -0.555570233f, -0.707106781f, -0.831469612f, -0.923879533f, -0.98078528f};
    constexpr float S[16] = {0.0f, -0.195090322f, -0.382683432f, -0.555570233f, -0.707106781f, -0.831469612f, -0.923879533f, -0.98078528f, -1.0f, -0.98078528f, -0.923879533f, -0.831469612f, -0.707106781f, -0.555570233f, -0.382683432f, -0.195090322f};
#pragma unroll
    for (int blk = 0; blk < 32; blk += 2 * HALF) {
#pragma unroll
        for (int j = 0; j < HALF; ++j) {
            const int a = brev5(blk + j), b = brev5(blk + j + HALF);       // registers of the two flow-graph positions
            constexpr int STEP = 16 / HALF;
            const int tw = j * STEP;                                        // w = W_32^tw = C + i S
            const float ar = re[a], ai = im[a], br = re[b], bi = im[b];
            if (tw == 0) {
                re[a] = ar + br;
                im[a] = ai + bi;
                re[b] = ar - br;
                im[b] = ai - bi;
            } else if (tw == 8) {                                           // w = -i: w b = (bi, -br)
                re[a] = ar + bi;
                im[a] = ai - br;
                re[b] = ar - bi;
                im[b] = ai + br;
            } else {
                const float pr = fmaf(bi, -S[tw], fmaf(br, C[tw], ar));     // Re(a + w b)
                const float pi = fmaf(bi, C[tw], fmaf(br, S[tw], ai));      // Im(a + w b)
                re[a] = pr;
                im[a] = pi;
                re[b] = fmaf(2.0f, ar, -pr);                                // a - w b = 2a - (a + w b)
                im[b] = fmaf(2.0f, ai, -pi);
            }
        }
    }
}

__device__ __forceinline__ void fft32_dif(float (&re)[32], float (&im)[32]) {
#if LEAF_FFT32_DIT
    fft32_dit_stage<1>(re, im);
    fft32_dit_stage<2>(re, im);
    fft32_dit_stage<4>(re, im);
    fft32_dit_stage<8>(re, im);
    fft32_dit_stage<16>(re, im);
#else
    fft32_stage<16>(re, im);
    fft32_stage<8>(re, im);
    fft32_stage<4>(re, im);
    fft32_stage<2>(re, im);
    fft32_stage<1>(re, im);
#endif
}

// Twiddle tables of the wave-level FFT, built once per workgroup in LDS:
//   twl[k1][lane] = W_2048^{lane*k1} = (cos, -sin)(2 pi lane k1 / 2048)   -- read with an immediate offset per k1
//   twh[j][h]     = h ? W_64^j : 1                                        -- the odd half-wave's twiddle of the cross stage
constexpr int kTwFloats = 2 * (32 * 64 + 32 * 2);
__device__ __forceinline__ void fft_build_twiddles(float2* twl, float2* twh, int tid, int nthreads) {
    for (int i = tid; i < 32 * 64; i += nthreads) {
        const int k1 = i >> 6, l = i & 63;
        float s, c;
        sincospif(2.0f * (float)((l * k1) & (kFftN - 1)) / (float)kFftN, &s, &c);
        twl[i] = make_float2(c, -s);
    }
    for (int i = tid; i < 64; i += nthreads) {
        const int j = i >> 1, h = i & 1;
        float s, c;
        sincospif(2.0f * (float)j / 64.0f, &s, &c);
        twh[i] = h ? make_float2(c, -s) : make_float2(1.0f, 0.0f);
    }
}

// Forward 2048-point DFT across one wave.  In: register r, lane l = element 64 r + l.  Out: register i, lane l =
// element 64 brev5(i) + l.  scr: wave-private LDS, >= 32*65 floats.
__device__ __forceinline__ void fft2048(float (&re)[32], float (&im)[32], float* scr, const float2* twl, const float2* twh,
                                        int lane) {
    fft32_dif(re, im);                                   // register i <-> k1 = brev5(i), lane = n2
#pragma unroll
    for (int i = 1; i < 32; ++i) {
        const float2 w = twl[brev5(i) * 64 + lane];
        const float r = re[i] * w.x - im[i] * w.y;
        im[i] = re[i] * w.y + im[i] * w.x;
        re[i] = r;
    }
    const int k1r = lane & 31, h = lane >> 5;
    float tr[32], ti[32];
#pragma unroll
    for (int i = 0; i < 32; ++i) scr[brev5(i) * 65 + lane] = re[i];
#pragma unroll
    for (int j = 0; j < 32; ++j) tr[j] = scr[k1r * 65 + j + 32 * h];
#pragma unroll
    for (int i = 0; i < 32; ++i) scr[brev5(i) * 65 + lane] = im[i];
#pragma unroll
    for (int j = 0; j < 32; ++j) ti[j] = scr[k1r * 65 + j + 32 * h];
    // 64-point DFT over n2 = j + 32 hh.  Radix-2 across the half-waves: the lower half needs a + b, the upper half
    // (a - b) * W64^j, where a / b are the lower / upper half-wave's values of the same register.  Two registers at a
    // time: v_permlane32_swap (gfx950: swaps the upper half of one VGPR with the lower half of another, in the VALU --
    // no LDS round trip) gathers [a_j | a_j'] and [b_j | b_j'], one add and one subtract serve both halves, and a second
    // swap puts [a+b | a-b] back in place.  Then the per-lane twiddle (1 on the lower half) and 32 points over j.
#if LEAF_FFT_SWAP
    auto cross = [](float& x0, float& x1) {
        auto g = __builtin_amdgcn_permlane32_swap(__float_as_uint(x0), __float_as_uint(x1), false, false);
        const float a = __uint_as_float(g[0]), b = __uint_as_float(g[1]);            // [a_j | a_j'], [b_j | b_j']
        auto q = __builtin_amdgcn_permlane32_swap(__float_as_uint(a + b), __float_as_uint(a - b), false, false);
        x0 = __uint_as_float(q[0]);                                                  // [a_j + b_j | a_j - b_j]
        x1 = __uint_as_float(q[1]);
    };
#pragma unroll
    for (int j = 0; j < 32; j += 2) {
        cross(tr[j], tr[j + 1]);
        cross(ti[j], ti[j + 1]);
    }
#pragma unroll
    for (int j = 0; j < 32; ++j) {
        if (j == 0) {
            re[j] = tr[j];
            im[j] = ti[j];
        } else {
            const float2 w = twh[2 * j + h];
            re[j] = tr[j] * w.x - ti[j] * w.y;
            im[j] = tr[j] * w.y + ti[j] * w.x;
        }
    }
#else
    const float sgn = h ? -1.0f : 1.0f;
    const int paddr = (lane ^ 32) << 2;
#pragma unroll
    for (int j = 0; j < 32; ++j) {
        const float pr = __int_as_float(__builtin_amdgcn_ds_bpermute(paddr, __float_as_int(tr[j])));
        const float pi = __int_as_float(__builtin_amdgcn_ds_bpermute(paddr, __float_as_int(ti[j])));
        const float ur = fmaf(tr[j], sgn, pr), ui = fmaf(ti[j], sgn, pi);
        if (j == 0) {
            re[j] = ur;
            im[j] = ui;
        } else {
            const float2 w = twh[2 * j + h];
            re[j] = ur * w.x - ui * w.y;
            im[j] = ur * w.y + ui * w.x;
        }
    }
#endif
    fft32_dif(re, im);                                   // register i <-> k' = brev5(i): element 64 k' + lane
}

// Pin a whole 32-register array at this point of the instruction stream: everything that produces it is scheduled above,
// everything after the statement below.  (Without it the compiler starts the next phase's loads under the last stage of a
// transform, runs out of registers and spills each loaded value behind a full vmcnt(0) wait.)
__device__ __forceinline__ void pin32(float (&a)[32]) {
    asm volatile("" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]),
                      "+v"(a[8]), "+v"(a[9]), "+v"(a[10]), "+v"(a[11]), "+v"(a[12]), "+v"(a[13]), "+v"(a[14]), "+v"(a[15])
                 : : "memory");
    asm volatile("" : "+v"(a[16]), "+v"(a[17]), "+v"(a[18]), "+v"(a[19]), "+v"(a[20]), "+v"(a[21]), "+v"(a[22]), "+v"(a[23]),
                      "+v"(a[24]), "+v"(a[25]), "+v"(a[26]), "+v"(a[27]), "+v"(a[28]), "+v"(a[29]), "+v"(a[30]), "+v"(a[31])
                 : : "memory");
}

// Eight samples of a clip around a block, for the time-domain terms of the even-window kernels:
// xa[j] = x[nb + 64 row(j) + lane], zero outside [0, T).  `interior` (wave-uniform: nb >= 0 and nb + 2048 <= T) takes
// loads at immediate offsets from one address; a clip's first and last blocks clamp and mask per lane (8 more
// instructions per sample).  The opaque offset holds the loads at this point of the stream, and each loaded value is pinned
// so that the mask stays a select and never becomes a branch round the load.
template <typename RowFn>
__device__ __forceinline__ void block_rows8(const void* x, size_t clip, int io_bf16, int nb, int T, int lane, bool interior,
                                            RowFn row, float (&xa)[8]) {
    const float* xb = static_cast<const float*>(x) + clip;
    const unsigned short* xh = static_cast<const unsigned short*>(x) + clip;
    int ofs = 0;
    asm volatile("" : "+v"(ofs) : : "memory");
    auto pin8 = [&]() {
        asm volatile("" : "+v"(xa[0]), "+v"(xa[1]), "+v"(xa[2]), "+v"(xa[3]), "+v"(xa[4]), "+v"(xa[5]), "+v"(xa[6]), "+v"(xa[7]));
    };
    // the four variants (sample type x interior / edge) are whole loops under wave-uniform branches: eight loads in flight
    if (interior) {
        const int n0 = nb + lane + ofs;
        if (io_bf16) {
#pragma unroll
            for (int j = 0; j < 8; ++j) xa[j] = __uint_as_float((unsigned)xh[n0 + 64 * row(j)] << 16);
        } else {
#pragma unroll
            for (int j = 0; j < 8; ++j) xa[j] = xb[n0 + 64 * row(j)];
        }
        pin8();
    } else {
        int nc[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) nc[j] = min(max(nb + 64 * row(j) + lane + ofs, 0), T - 1);
        if (io_bf16) {
#pragma unroll
            for (int j = 0; j < 8; ++j) xa[j] = __uint_as_float((unsigned)xh[nc[j]] << 16);
        } else {
#pragma unroll
            for (int j = 0; j < 8; ++j) xa[j] = xb[nc[j]];
        }
        pin8();
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int n = nb + 64 * row(j) + lane;
            xa[j] = (n >= 0 && n < T) ? xa[j] : 0.0f;
        }
    }
    asm volatile("" ::: "memory");
}

// Wave-wide sum, result in every lane; DPP inside the 16-lane rows, register swaps across them (no LDS).
__device__ __forceinline__ float wave_sum(float v) {
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0xb1, 0xf, 0xf, false));    // quad_perm [1,0,3,2]
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x4e, 0xf, 0xf, false));    // quad_perm [2,3,0,1]
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x124, 0xf, 0xf, false));   // row_ror:4
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x128, 0xf, 0xf, false));   // row_ror:8
    auto a = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    v = __uint_as_float(a[0]) + __uint_as_float(a[1]);
    auto b = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return __uint_as_float(b[0]) + __uint_as_float(b[1]);
}

// Wave-wide sums of four per-lane values at once: afterwards every lane of 16-lane row q holds the total of a_q.
__device__ __forceinline__ float wave_sum4_rows(float a0, float a1, float a2, float a3) {
    auto g = __builtin_amdgcn_permlane32_swap(__float_as_uint(a0), __float_as_uint(a2), false, false);
    const float b0 = __uint_as_float(g[0]) + __uint_as_float(g[1]);       // [a0 | a2] summed over the half-waves
    auto h = __builtin_amdgcn_permlane32_swap(__float_as_uint(a1), __float_as_uint(a3), false, false);
    const float b1 = __uint_as_float(h[0]) + __uint_as_float(h[1]);       // [a1 | a3]
    auto k = __builtin_amdgcn_permlane16_swap(__float_as_uint(b0), __float_as_uint(b1), false, false);
    float v = __uint_as_float(k[0]) + __uint_as_float(k[1]);              // rows [a0 a1 a2 a3], summed over row pairs
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0xb1, 0xf, 0xf, false));    // quad_perm [1,0,3,2]
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x4e, 0xf, 0xf, false));    // quad_perm [2,3,0,1]
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x124, 0xf, 0xf, false));   // row_ror:4
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x128, 0xf, 0xf, false));   // row_ror:8
    return v;
}

// Halving butterfly over 16 per-lane accumulators (acc[fi] = this lane's share of frame fi): afterwards every lane holds
// the wave-wide total of frame  fi(lane) = 8 b5 + 4 b4 + 2 b3 + b2  (b_k = bit k of the lane id).  No LDS: the two
// widest exchanges are gfx950 register swaps (v_permlane32_swap / v_permlane16_swap: swap the upper half / odd 16-lane
// rows of one VGPR with the lower half / even rows of another, so one add finishes both halves), the rest are DPP moves
// inside a 16-lane row (row_ror:8 = lane^8; row_shl/shr:4 under bank masks = lane^4; quad_perm = lane^2, lane^1).
__device__ __forceinline__ float frame_butterfly16(float (&acc)[16], int lane) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        auto g = __builtin_amdgcn_permlane32_swap(__float_as_uint(acc[i]), __float_as_uint(acc[i + 8]), false, false);
        acc[i] = __uint_as_float(g[0]) + __uint_as_float(g[1]);           // [frame i | frame i+8]
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        auto g = __builtin_amdgcn_permlane16_swap(__float_as_uint(acc[i]), __float_as_uint(acc[i + 4]), false, false);
        acc[i] = __uint_as_float(g[0]) + __uint_as_float(g[1]);           // rows: [i | i+4 | i+8 | i+12]
    }
    const bool up8 = (lane & 8) != 0, up4 = (lane & 4) != 0;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const float send = up8 ? acc[i] : acc[i + 2], keep = up8 ? acc[i + 2] : acc[i];
        acc[i] = keep + __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(send), 0x128, 0xf, 0xf, false));   // row_ror:8
    }
    {
        const float send = up4 ? acc[0] : acc[1], keep = up4 ? acc[1] : acc[0];
        int t = __builtin_amdgcn_update_dpp(0, __float_as_int(send), 0x104, 0xf, 0x5, false);   // row_shl:4 -> banks 0, 2
        t = __builtin_amdgcn_update_dpp(t, __float_as_int(send), 0x114, 0xf, 0xa, false);       // row_shr:4 -> banks 1, 3
        acc[0] = keep + __int_as_float(t);
    }
    float v = acc[0];
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x4e, 0xf, 0xf, false));   // quad_perm [2,3,0,1]
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0xb1, 0xf, 0xf, false));   // quad_perm [1,0,3,2]
    return v;
}

// Generic-geometry pooling of one frame: this lane's share of sum_n g[n - i_start] e[n] over the window's 64-sample
// rows r0 .. r0 + nt4 - 1.  The energies live in registers, so the row index has to be a compile-time constant: R is the
// template parameter the caller reaches through a switch on the (wave-uniform) first row; ge points at the pooling-row
// entry of row R for this lane, later rows are immediate offsets.  Rows past the window read the table's zero padding.
constexpr int kPoolRowsMax = 20;            // NT = ceil((K+63)/64) <= 17 for K <= 1025, rounded up to 4
template <int R>
__device__ __forceinline__ float pool_rows_from(const float (&er)[32], const float* ge, int nt4) {
    float acc = 0.0f;
#pragma unroll
    for (int t0 = 0; t0 < kPoolRowsMax; t0 += 4) {
        if (t0 < nt4 && R + t0 < 32) {
#pragma unroll
            for (int t = t0; t < t0 + 4; ++t)
                if (R + t < 32) acc = fmaf(er[R + t], ge[64 * t], acc);
        }
    }
    return acc;
}

// Generic-geometry backward of the pooling for one frame (transposed pooling): de[r] += gpre_m g[.] on the window's
// rows, and this lane's share of sum_n e[n] gpre_m g[.] (j - c)^2 (for d pool_w), with e recomputed from u = conj(y).
// Same compile-time-row-index scheme as pool_rows_from.
template <int R>
__device__ __forceinline__ void depool_rows_into(float (&de)[32], float& dpw, const float (&zre)[32], const float (&zim)[32],
                                                 float gpm, const float* ge, float tj0, int nt4, int lane, int Lv) {
#pragma unroll
    for (int t0 = 0; t0 < kPoolRowsMax; t0 += 4) {
        if (t0 < nt4 && R + t0 < 32) {
#pragma unroll
            for (int t = t0; t < t0 + 4; ++t)
                if (R + t < 32) {
                    const int i = brev5(R + t);                                // register holding row R + t of u
                    const float gw = gpm * ge[64 * t];
                    const float tj = tj0 + (float)(64 * t);                     // window position - centre, this lane
                    de[R + t] += gw;
                    const float e = 64 * (R + t) + lane < Lv ? zre[i] * zre[i] + zim[i] * zim[i] : 0.0f;   // as the forward masks it
                    dpw = fmaf(e * gw, tj * tj, dpw);
                }
        }
    }
}

// Block length (valid outputs per 2048-sample block).  Generic geometry: the largest multiple of 64 that fits.  Static
// pooling instances (compile-time K, hop) need every block to start on a frame boundary as well: the largest multiple
// of lcm(64, hop).
constexpr int fft_gcd(int a, int b) { return b == 0 ? a : fft_gcd(b, a % b); }
constexpr int fft_block_len(int K, int hop, bool stat) {
    const int unit = stat ? 64 / fft_gcd(64, hop) * hop : 64;
    return (kFftN - K + 1) / unit * unit;
}
// geometries with a static-pooling instantiation of leaf_fft_kernel (forward and backward)
constexpr bool fft_static_geometry(int K, int hop) {
    return (K == 401 && hop == 160) || (K == 801 && hop == 320) || (K == 201 && hop == 80);     // 16 / 32 / 8 kHz LEAF
}

// ---- spectra and pooling rows for the FFT path -------------------------------------------------------------
// One wave per filter: H[f][k] = (1/N) sum_j w_f[j] e^{+2 pi i jk/N} = conj(DFT(conj(w_f)))[k] / N, computed with the same
// wave-level fft2048 (the 1/N of the inverse transform is folded in); w_f = the taps exactly as
// convolution.py:88-90 hands them to conv1d (no tap cut here).
// Gz[f][kGPad + j] = g_f[j] (impulse_responses.py:74-80), zero elsewhere;  col_of[f] = f.
// One workgroup per filter: all waves evaluate the taps (into LDS), the pooling row and the twiddle tables; wave 0 then
// runs the transform.
constexpr int kPrepWaves = 8;
// Part A (every thread of the workgroup): conj(taps) of filter f into s_taps, the pooling row (global, and an LDS copy of the
// window when g_lds != NULL), the twiddle tables.  The caller synchronises before part B.
__device__ __forceinline__ void fft_prep_front(const float* __restrict__ kernel, const float* __restrict__ pool_w, int F, int K, int GZ,
                                               GaborBounds bd, float* __restrict__ Gz, int f, int which, float2* s_twl, float2* s_twh,
                                               float2* s_taps, float* g_lds, int tid) {
    const float mu = kernel[2 * f], sg = kernel[2 * f + 1];
    const float sgc = fminf(fmaxf(sg, bd.sigma_lo), bd.sigma_hi);
    for (int j = tid; j < K; j += kPrepWaves * 64) {
        float a, b;
        const float t = (float)(j - K / 2);
        gabor_tap(mu, sg, bd, t, a, b);
        if (which == 1) {
            const float a0 = a;
            a = -t * b;
            b = t * a0;
        } else if (which == 2) {
            const float c = t * t / (sgc * sgc * sgc) - 1.0f / sgc;
            a *= c;
            b *= c;
        }
        s_taps[j] = make_float2(a, -b);                   // conj(w)
    }
    if (which == 0) {                                     // pooling window row
        const float half = 0.5f * (float)(K - 1);
        for (int jj = tid; jj < GZ; jj += kPrepWaves * 64) {
            const int j = jj - kGPad;
            float v = 0.0f;
            if (j >= 0 && j < K) {
                const float q = ((float)j - half) / (pool_sigma(pool_w[f], K) * half);
                v = expf(-0.5f * (q * q));
                if (g_lds) g_lds[j] = v;
            }
            Gz[(size_t)f * GZ + jj] = v;
        }
    }
#if !defined(LEAF_PREP_ABLATE) || !(LEAF_PREP_ABLATE & 32)
    fft_build_twiddles(s_twl, s_twh, tid, kPrepWaves * 64);
#endif
}
// Part B (one wave): the filter's spectrum through the wave-level transform, stored; r_lds (optional): |R| of the
// real-spectrum form for the band-class decision (leaf_band.hpp).
__device__ __forceinline__ void fft_prep_transform(int F, int K, int real_spec, float2* __restrict__ H, int* __restrict__ col_of,
                                                   float* __restrict__ lone, int f, int which, const float2* s_twl, const float2* s_twh,
                                                   float* s_scr, const float2* s_taps, float* r_lds, int lane) {
    // real_spec (odd K): taps laid out zero-phase -- tap j sits at index (j - K/2) mod N -- so that the Hermitian
    // symmetry about the centre tap makes the spectrum real; the kernel rotates its input block to match.
    float re[32], im[32];
#pragma unroll
    for (int r = 0; r < 32; ++r) {
        const int i = 64 * r + lane;
        const int j = real_spec ? (i < kFftN / 2 ? i : i - kFftN) + K / 2 : i;
        re[r] = im[r] = 0.0f;
        // even K in real-spectrum form: the taps t = -(K/2 - 1) .. K/2 - 1 are Hermitian about t = 0; the unpaired tap
        // t = -K/2 (j = 0) is left out here and applied in the time domain by the kernel (lone tap, below)
        if (j >= (real_spec && !(K & 1) ? 1 : 0) && j < K) {
            const float2 t = s_taps[j];
            re[r] = t.x;
            im[r] = t.y;
        }
    }
#if !defined(LEAF_PREP_ABLATE) || !(LEAF_PREP_ABLATE & 16)
    fft2048(re, im, s_scr, s_twl, s_twh, lane);
#endif
    if (real_spec) {                                  // imaginary parts are rounding noise of exactly-cancelling pairs
        float* R = reinterpret_cast<float*>(H) + (size_t)which * F * kFftN;
#pragma unroll
        for (int i = 0; i < 32; ++i) {
            const float v = re[i] * (1.0f / kFftN);
            R[(size_t)f * kFftN + 64 * brev5(i) + lane] = v;
            if (r_lds) r_lds[64 * brev5(i) + lane] = fabsf(v);
        }
        if (lone && lane == 0) {                         // (even K) the unpaired tap w[t = -K/2] or its mu / sigma derivative
            const float2 c = s_taps[0];                   // conj(w)
            lone[((size_t)which * F + f) * 2] = c.x;
            lone[((size_t)which * F + f) * 2 + 1] = -c.y;
        }
    } else {
#pragma unroll
        for (int i = 0; i < 32; ++i)
            H[(size_t)f * kFftN + 64 * brev5(i) + lane] = make_float2(re[i] * (1.0f / kFftN), -im[i] * (1.0f / kFftN));
    }
    if (lane == 0 && which == 0) col_of[f] = f;
}
#ifndef LEAF_INST_TU               // non-template kernel: compiled once, in leaf_kernels.hip
__global__ __launch_bounds__(kPrepWaves * 64) void fft_prep_kernel(const float* __restrict__ kernel,
                                                                   const float* __restrict__ pool_w, int F, int K, int GZ,
                                                                   GaborBounds bd, int real_spec, float2* __restrict__ H,
                                                                   float* __restrict__ Gz, int* __restrict__ col_of,
                                                                   float* __restrict__ lone) {
    __shared__ float2 s_twl[32 * 64];
    __shared__ float2 s_twh[64];
    __shared__ float s_scr[32 * 65];
    __shared__ float2 s_taps[kFftN / 2 + 64];            // conj(w_f), K <= N/2 + 1
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int f = blockIdx.x;
    // blockIdx.y (backward tables, real-spectrum form only): 0 the taps w, 1 d w/d mu = i t w, 2 d w/d sigma =
    // (t^2/s^3 - 1/s) w (impulse_responses.py:5-16 differentiated; both stay Hermitian, so their spectra are real too)
    const int which = blockIdx.y;
    fft_prep_front(kernel, pool_w, F, K, GZ, bd, Gz, f, which, s_twl, s_twh, s_taps, nullptr, tid);
    __syncthreads();
    if (wave == 0) fft_prep_transform(F, K, real_spec, H, col_of, lone, f, which, s_twl, s_twh, s_scr, s_taps, nullptr, lane);
}
#endif

// Contiguous dealing of the nblocks = B * nblk blocks to G workgroups (the first `rem` get one more): workgroup w walks
// blocks [start(w), start(w) + count(w)).  A clip whose nblk blocks all fall to one workgroup is OWNED by it: its partial
// sums never leave that CU's reach and the workgroup finalizes it in its tail.  nblocks = 0: nothing is owned (kernels that
// do not finalize).
struct OwnedClips {
    int nblocks, G, nblk;
    __host__ __device__ int per() const { return nblocks / G; }
    __host__ __device__ int rem() const { return nblocks % G; }
    __host__ __device__ int start(int w) const { return w * per() + (w < rem() ? w : rem()); }
    __host__ __device__ int count(int w) const { return per() + (w < rem() ? 1 : 0); }
    __host__ __device__ int wg_of(int gb) const {
        const int q = per(), r = rem(), cut = r * (q + 1);
        return gb < cut ? gb / (q + 1) : r + (gb - cut) / (q > 0 ? q : 1);
    }
    __host__ __device__ bool owned(int b) const {
        return nblocks > 0 && wg_of(b * nblk) == wg_of(b * nblk + nblk - 1);
    }
};
// what the row arithmetic of the finalize step needs (fft_finalize_rows below)
struct FinParams {
    const float* part;     // [B][F][nslot][T']
    int F, TP;
    SlotGeom geo;
    const float *bias, *alpha, *delta, *root, *ema_w;
    float floor_;
    int mode;              // bit0 PCEN, bit1 log1p, bit2 bf16 output, bit3 no floor (the backward's raw pooled tensor)
    void* out;
    float* raw_out;
    const float* clip_scale2;   // LEAF_FLAG_PEAKNORM: [B] s_b^2 multiplying the pooled energies of clip b (NULL: none)
    // set by a workgroup kernel for its own tail only: the per-frame sums of the clips it owns sit in its LDS, already added
    // up ([rows from lds_row0][T']), instead of in `part`
    const float* lds_sums = nullptr;
    int lds_row0 = 0;
    const void* lds_coef = nullptr;   // (the same kernels) FinCoef[F] in its LDS, evaluated at the kernel's start: the tail does not wait for the parameters
};

// ---- band-limited filter tasks (leaf_band.hpp; static workgroup kernels, frame sums in LDS) ---------------------------------
// A frame whose window -- widened by the tails of the decimated pooling window -- is cut by the clip's ends (or would reach
// past them) is an EDGE frame: its sum over block `c` comes from a dense per-(filter, block) table; [lo, hi) is the part of
// the (cut) window that block owns, in absolute samples.
constexpr int kBandMaxEdge = 12;
struct BandEdge { int c, m, lo, hi; };
struct BandParams {
    const int* rec;        // [F][4]: bit 0 / 1 = the filter passes the 256- / 512-point criteria, first bin of its 256- / 512-point window (NULL: every filter on 2048 points)
    const float* gz;       // [F][kBandGzFloats]: decimated pooling windows of both classes
    const float* edge;     // [F][2][kBandMaxEdge][512]: edge-frame tables, register order of the class
    const float* gz2;      // (backward kernels) gz and edge built from the window g[j] (j - c)^2: d pool_w at the decimated rate
    const float* edge2;
    const int* elist;      // [kBandMaxEdge][4]: the edge list (c, m, lo, hi) in device memory
    int lds_off;           // float offset of the band area (plan, twiddle tables) in dynamic LDS
    int reg_lo, reg_hi;    // frames reg_lo .. reg_hi take the shift-invariant window, the others are edge frames
    int n_edge;
    const float* bias;     // (forward kernels) the pooling biases of THIS call, [F]: the energy bound of the class decision follows them
    float smax;            // > 1: the class decision may take them into account (leaf_band.hpp: band_bias_admits); NULL / <= 1: the strict decision alone (backward kernels, LEAF_ALGO_STRICT_BAND_CLASSES)
};

struct FftParams {
    const void* x;         // [B][T] fp32, or bf16 when io_bf16
    int io_bf16;
    const float2* H;       // [F][2048] complex spectra, or (real-spectrum kernels, odd K) [F][2048] floats
    const float* Gz;       // [F][GZ]
    float* part;           // [B][F][nslot][TP]: slot s = s-th block the frame's window meets (2, or 3 when K - 1 > L)
    int nslot;
    int B, T, TP, F, K, hop, padL;
    int L;                 // valid outputs per block
    int nblk;              // blocks per clip
    int GZ;                // row length of Gz (multiple of 4)
    int g_bufs;            // wave-private LDS pooling-row buffers: 2 (next row prefetched) when LDS allows, else 1
    int NT;                // 64-sample rows a pooling window can touch: ceil((K+63)/64)
    int scr_floats;        // wave-private LDS floats for transposes / energy rows
    int fq;                // filters per task: kFftFQ when the batch fills the chip, fewer (more, shorter tasks) when not
    int nfq;               // filter groups of fq
    int total_tasks;       // B * nblk * nfq  (one wave per task)
    // backward instantiation only (BWD = 1; real-spectrum tables R | R_mu | R_sigma stacked in H as [3][F][2048] floats)
    const float* gpre;     // [B][F][TP] dL/d(pooled pre-floor)
    const float* pool_w;   // [F] raw pooling widths
    float* dkpart;         // [B*nblk][F][2] per-block partial (d mu, d sigma), before the clamp sub-gradient
    float* dwpart;         // [B*nblk][F]    per-block partial d pool_w, before the clamp sub-gradient
    const float* lone;     // even K, real-spectrum form: [F][2] the unpaired tap w_f[t = -K/2] (backward: [3][F][2] with d/dmu, d/dsigma)
    int rot;               // rotation of the block for the real-spectrum form: K / 2 (= padL for odd K)
    unsigned long long* trace;   // LEAF_TRACE builds only
    // Clip-resident finalize (workgroup kernels): blocks are dealt contiguously, and with fin_fused != 0 a workgroup runs the
    // row arithmetic (bias, floor, EMA, PCEN) itself for every clip it owns outright; fft_finalize_kernel then handles only
    // the clips that straddle two workgroups.
    FinParams fin;
    int fin_fused;
    int stream_ring;       // STREAM kernels: frames of the LDS ring of per-frame partial sums (a power of two)
    BandParams band;       // band-limited filter tasks (rec == NULL: off)
    const float2* spec0;   // [workgroups][kWgRingFloat2]: the half spectrum of every workgroup's FIRST block, computed by the table
                           // launch on CUs it leaves idle (fft_prep_band_kernel); NULL: the kernel transforms it itself
};

constexpr unsigned leaf_layout_hash_fft() {                              // see leaf_layout_hash_fused (leaf_fused.hpp)
    unsigned h = leaf_layout_hash_fused();
    h = leaf_mix(h, sizeof(FftParams)); h = leaf_mix(h, offsetof(FftParams, part)); h = leaf_mix(h, offsetof(FftParams, lone));
    h = leaf_mix(h, offsetof(FftParams, trace)); h = leaf_mix(h, offsetof(FftParams, fin)); h = leaf_mix(h, offsetof(FftParams, stream_ring));
    h = leaf_mix(h, sizeof(FinParams)); h = leaf_mix(h, offsetof(FinParams, geo)); h = leaf_mix(h, offsetof(FinParams, out));
    h = leaf_mix(h, offsetof(FinParams, lds_row0)); h = leaf_mix(h, sizeof(OwnedClips)); h = leaf_mix(h, sizeof(SlotGeom));
    h = leaf_mix(h, offsetof(FftParams, band)); h = leaf_mix(h, sizeof(BandParams)); h = leaf_mix(h, offsetof(BandParams, n_edge));
    return h;
}

// SK/SHOP > 0: window and hop known at compile time (the reference's default 401/160): every frame/window offset of
// the pooling becomes an immediate, the energies never leave registers and no guard rows are needed.
// SK = 0: generic geometry, energies go through wave-private LDS rows.
//
// Every wave is independent (task = one block x one group of p.fq filters): no barrier after the twiddle tables are
// built, so the waves of a SIMD drift into different phases (register butterflies vs LDS transposes vs pooling) instead
// of colliding in lock step.  The filter spectrum H_f streams from L2 at the Z multiply (8-row chunks, one ahead); the
// pooling row g_f reaches wave-private LDS by 16-byte direct-to-LDS loads.  G2 = 1: two row buffers by filter parity,
// the next filter's row is requested right after this filter's Z multiply and lands under its transform and pooling;
// G2 = 0 (long windows, LDS too small for two rows per wave): one buffer, requested after the Z multiply, waited for
// after the transform.
//
// RS = 1 (odd K): real-spectrum form.  The reference's taps are exactly Hermitian about the window centre
// (w[-t] = conj(w[t]): a real even Gaussian times e^{i mu t}, impulse_responses.py:5-16), so with the taps laid out
// zero-phase (centre tap at index 0, negative times wrapped to the end of the block) their spectrum R_f is REAL.  The
// matching circular shift goes into the input instead: the block is loaded rotated by padL samples (a'[i] =
// a[(i + padL) mod N]), which is exactly A'[k] = A[k] e^{+i w_k padL}, and A H = A' R.  Per filter this halves the
// spectrum bytes (8 KB) and the multiply (2 instead of 4 ops per bin), and the 32 registers R_f needs are few enough
// to request the NEXT filter's spectrum during this filter's pooling -- which hides the L2 latency that otherwise
// stalls every filter (18 % of the kernel, tools/ablate_fft.py).  RS = 0 (even K: one unpaired tap breaks the
// symmetry): complex spectrum, loaded at the multiply.
template <int SK, int SHOP, int G2, int RS, int BWD>
__global__ __launch_bounds__(kFftWaves * 64, 2) void leaf_fft_kernel(const FftParams p) {
    extern __shared__ __attribute__((aligned(16))) float fsm2[];
    float2* twl = reinterpret_cast<float2*>(fsm2);                       // [32][64]
    float2* twh = twl + 32 * 64;                                          // [32][2]
    const int scr_floats = SK > 0 ? 32 * 65 : p.scr_floats;
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane0 = tid & 63;
    float* scr = reinterpret_cast<float*>(twh + 64) + (size_t)wave * (scr_floats + (G2 + 1) * p.GZ);
    float* sG = scr + scr_floats;                                         // [2][GZ] pooling rows, filter parity

    fft_build_twiddles(twl, twh, tid, kFftWaves * 64);
    __syncthreads();

    const int wave_global = blockIdx.x * kFftWaves + wave, wave_stride = gridDim.x * kFftWaves;
#if LEAF_TRACE
    int tr_n = 0;
#define FFT_STAMP()                                                                                   \
    do {                                                                                              \
        if (blockIdx.x == 0 && lane0 == 0 && tr_n < 64) p.trace[wave * 64 + tr_n] = __builtin_amdgcn_s_memtime(); \
        ++tr_n;                                                                                       \
    } while (0)
#else
#define FFT_STAMP() do { } while (0)
#endif
    for (int task = wave_global; task < p.total_tasks; task += wave_stride) {
        // The lane id is made opaque per task: everything derived from it that only the task prologue needs (row
        // offsets of the input block, address pairs) is then recomputed there instead of being hoisted out of the task
        // loop and spilled to scratch for the duration of the filter loop.
        int lane = lane0;
        asm volatile("" : "+v"(lane));
        FFT_STAMP();
        const int gb = task / p.nfq, fq = task - gb * p.nfq;
        const int b = gb / p.nblk, c = gb - b * p.nblk;
        const int n_c = c * p.L;
        const int Lv = min(p.L, p.T - n_c);
        const int f0 = fq * p.fq, f1 = min(p.F, f0 + p.fq);
        // pooling row of filter f -> wave-private LDS row buffer (f & g2), asynchronously
        constexpr int g2 = G2;                                             // 1: two row buffers (by filter parity), 0: one
        auto dma_pool_row = [&](int f) {
            asm volatile("" ::: "memory");
            const float* gsrc = p.Gz + (size_t)f * p.GZ;
            float* dst = sG + (f & g2) * p.GZ;
            // the static pooling reads [kGPad - 63, kGPad + K + 63) only; the generic one runs to the table's padded end.
            // 16 bytes per lane (gfx950 b128 LDS-DMA): 1 KB per instruction.
            if constexpr (SK > 0) {
                constexpr int GU = (kGPad + SK + 63 + 3) / 4 * 4;        // <= GZ by construction of the table
#pragma unroll
                for (int i0 = 0; i0 < GU; i0 += 256)
                    if (!(LEAF_FFT_ABLATE & 16) && (i0 + 256 <= GU || i0 + 4 * lane < GU))
                        __builtin_amdgcn_global_load_lds(gsrc + i0 + 4 * lane, (__attribute__((address_space(3))) void*)(dst + i0),
                                                         16, 0, 0);
            } else {
                for (int i0 = 0; i0 < p.GZ && !(LEAF_FFT_ABLATE & 16); i0 += 256)
                    if (i0 + 4 * lane < p.GZ)
                        __builtin_amdgcn_global_load_lds(gsrc + i0 + 4 * lane, (__attribute__((address_space(3))) void*)(dst + i0),
                                                         16, 0, 0);
            }
            asm volatile("" ::: "memory");
        };
        if (g2) dma_pool_row(f0);                                          // in flight under the forward transform
        // RS: rq[i] = R_f[64 brev5(i) + lane], the real spectrum row matching register i of the forward transform;
        // requested one phase ahead (here for the first filter, during the pooling for the following ones)
        float rq[RS ? 32 : 1];
        auto load_real_spectrum = [&](int f) {
            const float* src = reinterpret_cast<const float*>(p.H) + (size_t)((LEAF_FFT_ABLATE & 32) ? 0 : f) * kFftN + lane;
            asm volatile("" ::: "memory");
#pragma unroll
            for (int i = 0; i < (RS ? 32 : 0); ++i) rq[i] = (LEAF_FFT_ABLATE & 1) ? 1.0f + f : src[64 * brev5(i)];
            asm volatile("" ::: "memory");
        };
        if (RS && !BWD) load_real_spectrum(f0);
        // ---- spectrum of this block's input window (real input, imaginary part zero)
        float are[32], aim[32];
        {
            const float* xb = static_cast<const float*>(p.x) + (size_t)b * p.T;
            const unsigned short* xh = static_cast<const unsigned short*>(p.x) + (size_t)b * p.T;
            // bf16 input: every lane always loads (index clamped into the clip, value zeroed outside it), so that the 32
            // 2-byte loads are in flight together instead of one exec-masked load at a time (-9 % on 10 s clips)
            if (p.io_bf16) {
#pragma unroll
                for (int r = 0; r < 32; ++r) {
                    const int i = 64 * r + lane;                         // RS: block rotated left by padL samples
                    const int n = n_c - p.padL + (RS ? ((i + p.rot) & (kFftN - 1)) : i);
                    const unsigned v = xh[min(max(n, 0), p.T - 1)];
                    are[r] = (n >= 0 && n < p.T) ? __uint_as_float(v << 16) : 0.0f;
                    aim[r] = 0.0f;
                }
            } else {
#pragma unroll
                for (int r = 0; r < 32; ++r) {
                    const int i = 64 * r + lane;
                    const int n = n_c - p.padL + (RS ? ((i + p.rot) & (kFftN - 1)) : i);
                    are[r] = (n >= 0 && n < p.T) ? xb[n] : 0.0f;         // (the clamped form measured 7 % slower here)
                    aim[r] = 0.0f;
                }
            }
        }
        fft2048(are, aim, scr, twl, twh, lane);
        FFT_STAMP();
        int mlo = n_c + p.padL - p.K + 1;                                 // first frame whose window reaches the block
        mlo = mlo <= 0 ? 0 : (mlo + p.hop - 1) / p.hop;
        const int mhi = min(p.TP - 1, (n_c + Lv - 1 + p.padL) / p.hop);

        for (int f = f0; f < f1; ++f) {
            const float* sGf = sG + (f & g2) * p.GZ;                       // this filter's pooling row
            // ---- Z = conj(A * H) in natural register order, inverse transform by the conjugate trick.  The spectrum rows
            // come straight from L2 (register i of the forward transform <-> row brev5(i)) in four 8-row chunks, one chunk
            // requested ahead of the one being multiplied; the compiler barriers keep the loads from being hoisted into
            // one 64-register burst, which spills.
            float zre[32], zim[32];
            if constexpr (RS) {
                if (BWD) load_real_spectrum(f);                           // backward: no cross-filter prefetch (registers)
#pragma unroll
                for (int i = 0; i < 32; ++i) {
                    const int r = brev5(i);
                    zre[r] = are[i] * rq[i];
                    zim[r] = -(aim[i] * rq[i]);
                }
            } else {
                const float2* src = p.H + (size_t)((LEAF_FFT_ABLATE & 32) ? 0 : f) * kFftN + lane;
                float2 hq[2][8];
                auto load_chunk = [&](int c4) {
                    asm volatile("" ::: "memory");
#pragma unroll
                    for (int j = 0; j < 8; ++j)
                        hq[c4 & 1][j] = (LEAF_FFT_ABLATE & 1) ? make_float2(1.0f + f, 0.5f) : src[64 * brev5(8 * c4 + j)];
                    asm volatile("" ::: "memory");
                };
                load_chunk(0);
#pragma unroll
                for (int c4 = 0; c4 < 4; ++c4) {
                    if (c4 < 3) load_chunk(c4 + 1);
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        const int i = 8 * c4 + j, r = brev5(i);
                        const float2 h = hq[c4 & 1][j];
                        zre[r] = are[i] * h.x - aim[i] * h.y;
                        zim[r] = -(are[i] * h.y + aim[i] * h.x);
                    }
                }
            }
            // Pooling row DMA, issued only AFTER the last spectrum row has been consumed (the empty asm ties the issue point
            // to those products): vmcnt retires in issue order, so a DMA ahead of a spectrum wait is waited for with it
            // (measured: 9 % of the kernel).  With two row buffers it is the NEXT filter's row, in flight under this
            // filter's transform and pooling; it has retired before pooling f+1 because the spectrum loads of f+1, issued
            // after it, are waited for first.  With one buffer it is this filter's row, waited for below.
            asm volatile("" ::"v"(zre[brev5(31)]), "v"(zim[brev5(31)]) : "memory");
            if (g2) {
                if (f + 1 < f1) dma_pool_row(f + 1);
            } else {
                dma_pool_row(f);
            }
            FFT_STAMP();
            if (!(LEAF_FFT_ABLATE & 2)) fft2048(zre, zim, scr, twl, twh, lane);  // register i <-> samples 64 brev5(i) + lane
            if (!g2) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // single buffer: the row has landed in LDS
            // Even K, real-spectrum form: the Hermitian K - 1 taps went through the spectrum; the unpaired tap t = -K/2 is a
            // scaled copy of the input, y[cL + r] += w[-K/2] x[cL - padL + r].  In terms of u = conj(y) (register i <->
            // sample r = 64 brev5(i) + lane): u += conj(c) a[r].  The block samples come back from L1/L2 in 8-row chunks.
            const int nb_x = n_c - p.padL;                                // clip sample under the block's first sample
            const bool x_interior = nb_x >= 0 && nb_x + kFftN <= p.T;
            auto add_lone_tap = [&](float (&ure)[32], float (&uim)[32], float cre, float cim) {
                pin32(ure);                                               // the transform is complete before these loads issue
                pin32(uim);
#pragma unroll
                for (int i0 = 0; i0 < 32; i0 += 8) {
                    float xa[8];
                    block_rows8(p.x, (size_t)b * p.T, p.io_bf16, nb_x, p.T, lane, x_interior, [&](int j) { return brev5(i0 + j); }, xa);
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        ure[i0 + j] = fmaf(cre, xa[j], ure[i0 + j]);
                        uim[i0 + j] = fmaf(-cim, xa[j], uim[i0 + j]);
                    }
                    asm volatile("" : "+v"(ure[i0]), "+v"(ure[i0 + 1]), "+v"(ure[i0 + 2]), "+v"(ure[i0 + 3]), "+v"(ure[i0 + 4]),
                                      "+v"(ure[i0 + 5]), "+v"(ure[i0 + 6]), "+v"(ure[i0 + 7]), "+v"(uim[i0]), "+v"(uim[i0 + 1]),
                                      "+v"(uim[i0 + 2]), "+v"(uim[i0 + 3]), "+v"(uim[i0 + 4]), "+v"(uim[i0 + 5]), "+v"(uim[i0 + 6]),
                                      "+v"(uim[i0 + 7]));
                }
            };
            // backward tail shared by the static and generic instances: second transform of conj(dL/du), the two spectral
            // dot products and this (block, filter)'s partial gradients
            // (even K) lgr / lgi: this lane's share of dL/d(unpaired tap), real and imaginary part -- chained through the
            // tap's mu / sigma derivatives (p.lone[1], p.lone[2]) into the same two sums as the spectral part
            auto bwd_tail = [&](float (&vre)[32], float (&vim)[32], float dpw_over_c2, float lgr, float lgi) {
                fft2048(vre, vim, scr, twl, twh, lane);                  // g = dL/dS, register i <-> bin 64 brev5(i) + lane
                float amu = 0.0f, asg = 0.0f;
                if constexpr (RS == 2) {
                    const float* dmu = p.lone + ((size_t)p.F + f) * 2;
                    const float* dsg = p.lone + ((size_t)2 * p.F + f) * 2;
                    amu = dmu[0] * lgr + dmu[1] * lgi;
                    asg = dsg[0] * lgr + dsg[1] * lgi;
                }
                {
                    const float* rmu = reinterpret_cast<const float*>(p.H) + ((size_t)p.F + f) * kFftN + lane;
                    const float* rsg = reinterpret_cast<const float*>(p.H) + ((size_t)2 * p.F + f) * kFftN + lane;
#pragma unroll
                    for (int i0 = 0; i0 < 32; i0 += 8) {
                        float tm[8], ts[8];
                        asm volatile("" ::: "memory");
#pragma unroll
                        for (int j = 0; j < 8; ++j) {
                            tm[j] = rmu[64 * brev5(i0 + j)];
                            ts[j] = rsg[64 * brev5(i0 + j)];
                        }
                        asm volatile("" ::: "memory");
#pragma unroll
                        for (int j = 0; j < 8; ++j) {
                            const int i = i0 + j;
                            const float dR = are[i] * vre[i] + aim[i] * vim[i];
                            amu = fmaf(dR, tm[j], amu);
                            asg = fmaf(dR, ts[j], asg);
                        }
                    }
                }
                amu = wave_sum(amu);
                asg = wave_sum(asg);
                dpw_over_c2 = wave_sum(dpw_over_c2);
                if (lane == 0) {
                    const float sp = pool_sigma(p.pool_w[f], p.K);
                    p.dkpart[((size_t)gb * p.F + f) * 2] = amu;
                    p.dkpart[((size_t)gb * p.F + f) * 2 + 1] = asg;
                    p.dwpart[(size_t)gb * p.F + f] = dpw_over_c2 / (sp * sp * sp);
                }
            };
            FFT_STAMP();
            if constexpr (SK > 0) {
                // ---- static geometry: frame df (relative to the block's first hop) has its window at
                // i in [df*SHOP - padL, +SK); n_c is a multiple of SHOP, so all of this is compile-time.
                constexpr int PADL = SK / 2 + SK % 2 - 1;
                constexpr int LS = fft_block_len(SK, SHOP, true);
                static_assert(LS % SHOP == 0 && LS % 64 == 0 && LS > 0, "block length must be a whole number of hops and rows");
                constexpr int DMIN = -((SK - 1 - PADL) / SHOP);
                constexpr int DMAX = (LS - 1 + PADL) / SHOP;
                constexpr int NFR = DMAX - DMIN + 1;
                constexpr int NROW = LS / 64;
                static_assert(NFR <= 32, "at most two butterfly groups of 16 frames");
                constexpr int NGRP = (NFR + 15) / 16;
                if constexpr (BWD) {
                    // g_pre of the NFR frames this block meets, as wave-uniform scalars
                    float gp[NFR];
                    {
                        const int fi = lane & 31, m = n_c / SHOP + DMIN + fi;
                        const float mine = (fi < NFR && m >= mlo && m <= mhi) ? p.gpre[((size_t)b * p.F + f) * p.TP + m] : 0.0f;
#pragma unroll
                        for (int q = 0; q < NFR; ++q) gp[q] = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(mine), q));
                    }
                    constexpr float HALF = 0.5f * (float)(SK - 1);
                    const float lanef = (float)lane;
                    float dpw = 0.0f;
                    float vre[32], vim[32];                                  // conj(dL/du), natural row order
#pragma unroll
                    for (int r = 0; r < 32; ++r) {
                        const int i = brev5(r);                              // register holding row r of u
                        if (r < NROW) {
                            const float ur = zre[i], ui = zim[i];
                            const bool ok = 64 * r + lane < Lv;
                            float de = 0.0f, dq = 0.0f;
#pragma unroll
                            for (int fi = 0; fi < NFR; ++fi) {
                                const int is = (DMIN + fi) * SHOP - PADL;
                                if (is <= 64 * r + 63 && is + SK > 64 * r) {
                                    const float gw = gp[fi] * sGf[kGPad + 64 * r - is + lane];   // zero outside the window
                                    const float tj = (float)(64 * r - is) - HALF + lanef;           // window position - centre
                                    de += gw;
                                    dq = fmaf(gw, tj * tj, dq);
                                }
                            }
                            const float e = ok ? ur * ur + ui * ui : 0.0f;
                            dpw = fmaf(e, dq, dpw);
                            const float s2 = ok ? 2.0f * de : 0.0f;
                            vre[r] = s2 * ur;
                            vim[r] = -(s2 * ui);
                        } else {
                            vre[r] = vim[r] = 0.0f;                          // circular wrap-around outputs: no gradient
                        }
                    }
                    bwd_tail(vre, vim, dpw / (HALF * HALF), 0.0f, 0.0f);
                    FFT_STAMP();
                    continue;
                }
                float er[NROW];                                            // energies of the valid outputs, row r
#pragma unroll
                for (int i = 0; i < 32; ++i) {
                    const int r = brev5(i);
                    if (r < NROW) er[r] = 64 * r + lane < Lv ? zre[i] * zre[i] + zim[i] * zim[i] : 0.0f;
                }
                if (RS && f + 1 < f1) {                                   // Z is dead: next filter's spectrum, in flight
                    asm volatile("" ::"v"(er[0]), "v"(er[NROW - 1]));    // under the pooling and the reduction
                    load_real_spectrum(f + 1);
                }
                float acc[NGRP][16];                                       // frame fi -> acc[fi / 16][fi % 16]
#pragma unroll
                for (int g = 0; g < NGRP; ++g)
#pragma unroll
                    for (int fi = 0; fi < 16; ++fi) acc[g][fi] = 0.0f;
#pragma unroll
                for (int r = 0; r < NROW; ++r) {
#pragma unroll
                    for (int fi = 0; fi < NFR; ++fi) {
                        const int is = (DMIN + fi) * SHOP - PADL;
                        if (is <= 64 * r + 63 && is + SK > 64 * r) {
                            if (LEAF_FFT_ABLATE & 4) acc[fi / 16][fi % 16] += er[r];
                            else acc[fi / 16][fi % 16] = fmaf(er[r], sGf[kGPad + 64 * r - is + lane], acc[fi / 16][fi % 16]);
                        }
                    }
                }
                asm volatile("" : "+v"(acc[0][0]));
#pragma unroll
                for (int g = 0; g < NGRP; ++g) {
                    float v;
                    if (LEAF_FFT_ABLATE & 8) {
                        v = acc[g][0];
#pragma unroll
                        for (int i = 1; i < 16; ++i) v += acc[g][i];
                    } else {
                        v = frame_butterfly16(acc[g], lane);
                    }
                    const int fi = 16 * g + ((lane >> 5) & 1) * 8 + ((lane >> 4) & 1) * 4 + ((lane >> 3) & 1) * 2 + ((lane >> 2) & 1);
                    const int m = n_c / SHOP + DMIN + fi;
                    if ((lane & 3) == 0 && fi < NFR && m >= mlo && m <= mhi && (!(LEAF_FFT_ABLATE & 64) || v == 12345.678f)) {
                        const int first_block = max(0, m * p.hop - p.padL) / p.L;
                        p.part[(((size_t)b * p.F + f) * p.nslot + (c - first_block)) * p.TP + m] = v;
                    }
                }
                FFT_STAMP();
            } else {
                if constexpr (BWD) {
                    // ---- generic geometry, backward: transposed pooling into de[32] (registers), frame by frame
                    if constexpr (RS == 2) add_lone_tap(zre, zim, p.lone[2 * f], p.lone[2 * f + 1]);
                    float de[32];
#pragma unroll
                    for (int r = 0; r < 32; ++r) de[r] = 0.0f;
                    float dpw = 0.0f;
                    const float half = 0.5f * (float)(p.K - 1), lanef = (float)lane;
                    for (int m = mlo; m <= mhi; ++m) {
                        const float gpm = p.gpre[((size_t)b * p.F + f) * p.TP + m];
                        const int i_start = m * p.hop - p.padL - n_c;
                        const int r0 = i_start > 0 ? i_start >> 6 : 0;
                        const float* ge = sGf + kGPad + (64 * r0 - i_start) + lane;
                        const float tj0 = (float)(64 * r0 - i_start) - half + lanef;
                        const int nt4b = ((min(31, (i_start + p.K - 1) >> 6) - r0 + 1) + 3) & ~3;
                        switch (r0) {
#define LEAF_POOL_CASE(R) case R: depool_rows_into<R>(de, dpw, zre, zim, gpm, ge, tj0, nt4b, lane, Lv); break;
                            LEAF_POOL_CASE(0) LEAF_POOL_CASE(1) LEAF_POOL_CASE(2) LEAF_POOL_CASE(3) LEAF_POOL_CASE(4)
                            LEAF_POOL_CASE(5) LEAF_POOL_CASE(6) LEAF_POOL_CASE(7) LEAF_POOL_CASE(8) LEAF_POOL_CASE(9)
                            LEAF_POOL_CASE(10) LEAF_POOL_CASE(11) LEAF_POOL_CASE(12) LEAF_POOL_CASE(13) LEAF_POOL_CASE(14)
                            LEAF_POOL_CASE(15) LEAF_POOL_CASE(16) LEAF_POOL_CASE(17) LEAF_POOL_CASE(18) LEAF_POOL_CASE(19)
                            LEAF_POOL_CASE(20) LEAF_POOL_CASE(21) LEAF_POOL_CASE(22) LEAF_POOL_CASE(23) LEAF_POOL_CASE(24)
                            LEAF_POOL_CASE(25) LEAF_POOL_CASE(26) LEAF_POOL_CASE(27) LEAF_POOL_CASE(28) LEAF_POOL_CASE(29)
                            LEAF_POOL_CASE(30) LEAF_POOL_CASE(31)
#undef LEAF_POOL_CASE
                            default: break;
                        }
                    }
                    float vre[32], vim[32];                                  // conj(dL/du); no gradient past the clip's end
#pragma unroll
                    for (int r = 0; r < 32; ++r) {
                        const int i = brev5(r);
                        const bool ok = 64 * r + lane < Lv;
                        const float s2 = ok ? 2.0f * de[r] : 0.0f;
                        vre[r] = s2 * zre[i];
                        vim[r] = -(s2 * zim[i]);
                    }
                    float lgr = 0.0f, lgi = 0.0f;
                    if constexpr (RS == 2) {
                        // u = u_H + conj(c) x  =>  dL/dc_re = sum_n x[n] Re v[n], dL/dc_im = sum_n x[n] Im v[n].  Only v and the
                        // block spectrum are live here (de and u are dead): the samples come back in 8-row chunks.
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
                    }
                    bwd_tail(vre, vim, dpw / (half * half), lgr, lgi);
                    FFT_STAMP();
                    continue;
                }
                // ---- generic geometry: energies of the valid outputs stay in registers (natural row order)
                if constexpr (RS == 2) add_lone_tap(zre, zim, p.lone[2 * f], p.lone[2 * f + 1]);
                float er[32];
#pragma unroll
                for (int i = 0; i < 32; ++i) {
                    const int r = brev5(i);
                    er[r] = 64 * r + lane < Lv ? zre[i] * zre[i] + zim[i] * zim[i] : 0.0f;
                }
                if (RS && f + 1 < f1) load_real_spectrum(f + 1);         // Z is dead: in flight under the pooling
                // ---- Gaussian pooling of every frame whose window meets this block.  Per frame (wave-uniform loop): a
                // switch on the window's first row selects the unrolled row code (compile-time register indices,
                // immediate LDS offsets); four frames share one DPP/permlane reduction that leaves frame q's total in
                // 16-lane row q, and four lanes store four consecutive frames.
                for (int mg = mlo; mg <= mhi; mg += 4) {                  // four frames per pass, one combined reduction
                    float acc[4];
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        acc[j] = 0.0f;
                        const int m = mg + j;
                        if (m <= mhi) {
                            const int i_start = m * p.hop - p.padL - n_c;
                            const int r0 = i_start > 0 ? i_start >> 6 : 0;
                            const float* ge = sGf + kGPad + (64 * r0 - i_start) + lane;
                            // rows this window really touches, rounded up to the unrolling of 4 (the table keeps 256 zeros
                            // behind the window for the round-up)
                            const int nt4 = ((min(31, (i_start + p.K - 1) >> 6) - r0 + 1) + 3) & ~3;
                            switch (r0) {
#define LEAF_POOL_CASE(R) case R: acc[j] = pool_rows_from<R>(er, ge, nt4); break;
                                LEAF_POOL_CASE(0) LEAF_POOL_CASE(1) LEAF_POOL_CASE(2) LEAF_POOL_CASE(3) LEAF_POOL_CASE(4)
                                LEAF_POOL_CASE(5) LEAF_POOL_CASE(6) LEAF_POOL_CASE(7) LEAF_POOL_CASE(8) LEAF_POOL_CASE(9)
                                LEAF_POOL_CASE(10) LEAF_POOL_CASE(11) LEAF_POOL_CASE(12) LEAF_POOL_CASE(13) LEAF_POOL_CASE(14)
                                LEAF_POOL_CASE(15) LEAF_POOL_CASE(16) LEAF_POOL_CASE(17) LEAF_POOL_CASE(18) LEAF_POOL_CASE(19)
                                LEAF_POOL_CASE(20) LEAF_POOL_CASE(21) LEAF_POOL_CASE(22) LEAF_POOL_CASE(23) LEAF_POOL_CASE(24)
                                LEAF_POOL_CASE(25) LEAF_POOL_CASE(26) LEAF_POOL_CASE(27) LEAF_POOL_CASE(28) LEAF_POOL_CASE(29)
                                LEAF_POOL_CASE(30) LEAF_POOL_CASE(31)
#undef LEAF_POOL_CASE
                                default: break;
                            }
                        }
                    }
                    const float v = wave_sum4_rows(acc[0], acc[1], acc[2], acc[3]);   // row q of the wave: frame mg + q
                    const int m = mg + (lane >> 4);
                    if ((lane & 15) == 0 && m <= mhi) {
                        const int first_block = max(0, m * p.hop - p.padL) / p.L;
                        p.part[(((size_t)b * p.F + f) * p.nslot + (c - first_block)) * p.TP + m] = v;
                    }
                }
            }
        }
    }
}

// ---- finalize for the overlap-save path -----------------------------------------------------------------------
// part is [B][F][nslot][T'] (frame-contiguous).  Per frame: sum of the valid slots in block order (a frame's window meets
// one or two blocks, three when K - 1 > L: computed from the geometry, so the buffer needs no zero fill), x s_b^2 with
// LEAF_FLAG_PEAKNORM, + bias (pooling.py:41) -> floor (frontend.py:84) -> EMA (postprocessing.py:13-28) -> PCEN
// (postprocessing.py:62-69).  Mode bits: 1 PCEN, 2 log1p, 4 bf16 output, 8 no floor (the backward's raw pooled tensor).
//
// ONE arithmetic for every kernel that finalizes (round 3): the functions below.  The EMA is the reference's own
// recurrence, evaluated sequentially in its fp32 operation order -- acc = (w * x) + ((1 - w) * acc), acc_{-1} = x_0 -- so
// that a row can be finalized a few frames at a time as its blocks complete (the workgroup kernels' streaming finalize:
// no partial sums in HBM, no second kernel) and still be bit-identical to the row kernel below, which other batch sizes
// and the other kernel families use.  (Rounds 1-2 composed affine maps in a wavefront scan: same value to ~1e-7, but an
// association order that only exists for a whole 128-frame chunk.)
struct FinCoef {
    float bias, w, omw, a, inv_r, dl, d_r, inv_d;
};
__device__ __forceinline__ FinCoef fin_coef(const FinParams& q, int f) {
#pragma clang fp contract(off)
    FinCoef c{};
    c.bias = q.bias ? q.bias[f] : 0.0f;
    if (q.mode & 1) {
        c.w = fminf(fmaxf(q.ema_w[f], 0.0f), 1.0f);              // postprocessing.py:14
        c.omw = 1.0f - c.w;
        c.a = fminf(q.alpha[f], 1.0f);                            // postprocessing.py:63
        c.inv_r = 1.0f / fmaxf(q.root[f], 1.0f);                  // postprocessing.py:64,66
        c.dl = q.delta[f];
        c.d_r = c.dl > 0.0f ? leaf_pow_pos(c.dl, c.inv_r) : powf(c.dl, c.inv_r);
        c.inv_d = c.dl > 0.0f ? 1.0f / c.dl : 0.0f;
    }
    return c;
}
// slots a frame's window meets (1..nslot): blocks of L valid outputs, window [m hop - padL, m hop - padL + K)
__device__ __forceinline__ int fin_slots(const SlotGeom& geo, int m) {
    const int s0 = m * geo.hop - geo.padL;
    return min(geo.T - 1, s0 + geo.K - 1) / geo.L - max(0, s0) / geo.L + 1;
}
// pre-floor pooled value from the slot values (a = earliest block); s2 = the clip's LEAF_FLAG_PEAKNORM scale or 1
__device__ __forceinline__ float fin_pooled(float a, float b, float c3, int ns, bool scaled, float s2, float bias) {
#pragma clang fp contract(off)     // (HIP's __fmul_rn / __fadd_rn are plain operators: only the pragma stops an FMA from forming)
    float x = a;
    if (ns > 1) x = __fadd_rn(x, b);
    if (ns > 2) x = __fadd_rn(x, c3);
    if (scaled) x = __fmul_rn(x, s2);
    return __fadd_rn(x, bias);
}
// postprocessing.py:22, literally (no contraction: every kernel rounds the same way, and the way the reference does)
__device__ __forceinline__ float fin_ema_step(const FinCoef& c, float p, float M) {
#pragma clang fp contract(off)
    const float t1 = c.w * p, t2 = c.omw * M;
    return t1 + t2;
}
// the output value of one frame from its floored pooled value p and smoothed value M (unused without PCEN).
// q = p / (floor+M)^a with the hardware log2/exp2 (1 ulp each; floor+M is a normal number); then
// (q+d)^(1/r) - d^(1/r) = d^(1/r) expm1(log1p(q/d)/r) for d > 0 (no cancelling subtraction); for d <= 0 the reference's
// formula is followed literally.
__device__ __forceinline__ float fin_point(const FinCoef& c, int mode, float floor_, float p, float M) {
#pragma clang fp contract(off)     // the same bits from every kernel that finalizes (leaf_fastmath.hpp)
    if (mode & 1) {
        // the arguments of the inlined helpers are materialised first: a product feeding `1.0f + z` inside one of them would
        // otherwise be a contraction candidate across the call boundary
        const float qv = __fmul_rn(p, leaf_pow_pos(__fadd_rn(floor_, M), -c.a));
        if (c.dl > 0.0f) return __fmul_rn(c.d_r, leaf_expm1_pos(__fmul_rn(c.inv_r, leaf_log1p_pos(__fmul_rn(qv, c.inv_d)))));
        return powf(qv + c.dl, c.inv_r) - c.d_r;
    }
    return (mode & 2) ? log1pf(p) : p;
}
// the PCEN branch of fin_point for delta > 0, without a branch in it (same operations, same bits): for callers that have
// checked the sign for the whole wave and want several frames' chains interleaved
__device__ __forceinline__ float fin_point_pcen_pos(const FinCoef& c, float floor_, float p, float M) {
#pragma clang fp contract(off)
    const float qv = __fmul_rn(p, leaf_pow_pos(__fadd_rn(floor_, M), -c.a));
    return __fmul_rn(c.d_r, leaf_expm1_pos(__fmul_rn(c.inv_r, leaf_log1p_pos(__fmul_rn(qv, c.inv_d)))));
}
[[maybe_unused]] __device__ __noinline__ float fin_point_outofline(const FinCoef& c, int mode, float floor_, float p, float M) {
    return fin_point(c, mode, floor_, p, M);
}
__device__ __forceinline__ void fin_store(const FinParams& q, size_t o, float v) {
    if (q.mode & 4) {                                            // bf16 output, round to nearest even
        const unsigned u = __float_as_uint(v);
        static_cast<unsigned short*>(q.out)[o] = (unsigned short)((u + 0x7fffu + ((u >> 16) & 1u)) >> 16);
    } else {
        static_cast<float*>(q.out)[o] = v;
    }
}

// A tile of up to 64 rows x all T' frames, 64 frames at a time, by NTHREADS threads of one workgroup (all of them call;
// `tile` = kFinTileFloats floats of LDS):  (1) all threads: slots -> pooled -> floor into the P tile (coalesced along the
// frames of a row);  (2) wave 0, lane = row: the EMA recurrence down the 64 frames, state carried in a register between
// chunks, into the M tile;  (3) all threads: the PCEN point function and the store.  COHERENT: the caller's own workgroup
// wrote `part` a moment ago (its stores are in L2: release + barrier on the caller's side), so the partial sums are read
// past this CU's vector cache -- relaxed agent-scope loads (`sc1`), NOT an agent-scope acquire fence: on this multi-die
// part `buffer_inv sc1` also drops the XCD's L2 lines, which every other workgroup of the die is still using (measured:
// +15 us per launch).  `own` (row kernel only): rows of clips the main kernel already finalized are skipped.
// LDS floats of a tile: three P buffers and two M buffers of ROWS x (COLS + 1) (the chunks are software-pipelined), the
// rows' coefficients, scales and live flags
template <int ROWS, int COLS>
constexpr int fin_tile_floats() { return 5 * ROWS * (COLS + 1) + ROWS * 8 + ROWS * 2; }
template <int ROWS, int COLS>
constexpr int fin_tile_floats_single() { return 2 * ROWS * (COLS + 1) + ROWS * 8 + ROWS * 2; }   // rows of one chunk: T' <= COLS
// A tile of up to ROWS rows x all T' frames by the NTHREADS threads of one workgroup (all of them call; `tile` =
// fin_tile_floats floats of LDS), COLS frames per chunk, three stages per chunk:
//   (1) worker waves: slots -> pooled -> floor into a P buffer (coalesced along the frames of a row);
//   (2) wave 0, lane = row: the EMA recurrence down the chunk's frames, state carried in a register, into an M buffer;
//   (3) worker waves: the PCEN point function and the store.
// The stages of consecutive chunks overlap (stage 1 of chunk k + 1 and stage 3 of chunk k - 1 run beside stage 2 of chunk
// k; one barrier per step), so a long row costs about its recurrence -- T' dependent multiply-add pairs -- and a short one
// three dependent phases.  COHERENT: the caller's own workgroup wrote `part` a moment ago (its stores are in L2: release +
// barrier on the caller's side), so the partial sums are read past this CU's vector cache -- relaxed agent-scope loads
// (`sc1`), NOT an agent-scope acquire fence: on this multi-die part `buffer_inv sc1` also drops the XCD's L2 lines, which
// every other workgroup of the die is still using (measured: +15 us per launch).  `own` (row kernel only): rows of clips
// the main kernel already finalized are skipped.
template <bool COHERENT, int ROWS, int COLS>
__device__ __forceinline__ void fft_finalize_tile(const FinParams& q, int row0, int nrows, const OwnedClips& own, float* tile,
                                                  int tid, int nthreads) {
    static_assert((COLS & (COLS - 1)) == 0 && ROWS <= 64, "COLS a power of two; one lane of wave 0 per row");
    constexpr int STRIDE = COLS + 1;                             // + 1: column reads (lane = row) hit distinct banks
    const int NWORK = nthreads - 64;                             // wave 0 runs the recurrence, the others (>= 2 waves) stages 1 and 3
    // buffers interleaved P0 M0 P1 M1 P2 (then the coefficients): a row of one chunk (T' <= COLS) only touches P0 and M0, so
    // such a caller may hand over two buffers' worth of LDS: fin_tile_floats_single
    constexpr int BUF = ROWS * STRIDE;
    auto Pb = [&](int i) { return tile + (2 * i) * BUF; };       // i = 0..2
    auto Mb = [&](int i) { return tile + (2 * i + 1) * BUF; };   // i = 0..1
    const bool single = q.TP <= COLS;
    FinCoef* coef = reinterpret_cast<FinCoef*>(tile + (single ? 2 : 5) * BUF);
    float* s2row = reinterpret_cast<float*>(coef + ROWS);
    int* live = reinterpret_cast<int*>(s2row + ROWS);
    const int TP = q.TP, mode = q.mode;
    const SlotGeom geo = q.geo;
    auto ld = [](const float* a) {
        if constexpr (COHERENT) return __hip_atomic_load(a, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        else return *a;
    };
    if (tid < nrows) {
        const int row = row0 + tid, b = row / q.F;
        coef[tid] = q.lds_coef ? static_cast<const FinCoef*>(q.lds_coef)[row - b * q.F] : fin_coef(q, row - b * q.F);
        s2row[tid] = q.clip_scale2 ? q.clip_scale2[b] : 1.0f;
        live[tid] = own.nblocks > 0 && own.owned(b) ? 0 : 1;
    }
    __syncthreads();
    const bool scaled = q.clip_scale2 != nullptr;
    const bool worker = tid >= 64;
    const int wt = tid - 64;
    const int nchunk = (TP + COLS - 1) / COLS;
    const int nel = nrows * COLS;
    float carry = 0.0f;                                          // wave 0, lane = row: EMA state after the previous chunk
    for (int step = 0; step < nchunk + 2; ++step) {
        if (worker) {
            // steady state of a long row (stages 1 and 3 both have a chunk): the worker waves split into two halves, one per
            // stage, so that the two latency chains overlap; otherwise every worker runs the one stage there is
            const bool both = step < nchunk && step >= 2;
            const int NHALF = (NWORK / 128) * 64;                // threads of the first half (whole waves)
            const bool do1 = step < nchunk && (!both || wt < NHALF);
            const bool do3 = step >= 2 && (!both || wt >= NHALF);
            const int nw1 = both ? NHALF : NWORK, w1 = wt;
            const int nw3 = both ? NWORK - NHALF : NWORK, w3 = both ? wt - NHALF : wt;
            if (do1) {                                           // ---- stage 1 of chunk `step`
                const int m0 = step * COLS, nfr = min(COLS, TP - m0);
                float* P = Pb(step % 3);
                // four elements per thread and pass: their (up to twelve) loads are issued together from clamped, always
                // valid addresses and selected afterwards -- one memory round trip per pass instead of one per element
                constexpr int U = 4;
                for (int base = w1; base < nel; base += U * nw1) {
                    float a[U], b2[U], c3[U];
                    int ns[U];
#pragma unroll
                    for (int u = 0; u < U; ++u) {
                        const int idx = min(base + u * nw1, nel - 1);
                        const int r = idx / COLS, m = min(m0 + (idx & (COLS - 1)), TP - 1);
                        if (q.lds_sums) {                                 // the slots were added up in LDS (a + b: the same rounding)
                            ns[u] = 1;
                            a[u] = q.lds_sums[(size_t)(row0 + r - q.lds_row0) * TP + m];
                            b2[u] = c3[u] = 0.0f;
                            continue;
                        }
                        const float* pr = q.part + (size_t)(row0 + r) * geo.nslot * TP + m;
                        ns[u] = fin_slots(geo, m);
                        a[u] = ld(pr);
                        b2[u] = ld(pr + (ns[u] > 1 ? TP : 0));
                        c3[u] = geo.nslot > 2 ? ld(pr + (ns[u] > 2 ? 2 * TP : 0)) : 0.0f;
                    }
#pragma unroll
                    for (int u = 0; u < U; ++u) {
                        const int idx = base + u * nw1;
                        const int r = min(idx, nel - 1) / COLS, j = idx & (COLS - 1);
                        if (idx < nel && j < nfr && live[r]) {
                            float v = fin_pooled(a[u], b2[u], c3[u], ns[u], scaled, s2row[r], coef[r].bias);
                            if (q.raw_out) q.raw_out[(size_t)(row0 + r) * TP + m0 + j] = v;
                            if (!(mode & 8)) v = pooled_floor(v);
                            P[r * STRIDE + j] = v;
                        }
                    }
                }
            }
            if (do3) {                                           // ---- stage 3 of chunk `step - 2`
                const int k = step - 2, m0 = k * COLS, nfr = min(COLS, TP - m0);
                const float* P = Pb(k % 3);
                const float* Mt = Mb(k & 1);
                for (int idx = w3; idx < nel; idx += nw3) {
                    const int r = idx / COLS, j = idx & (COLS - 1);
                    if (j < nfr && live[r])
                        fin_store(q, (size_t)(row0 + r) * TP + m0 + j,
                                  fin_point(coef[r], mode, q.floor_, P[r * STRIDE + j], Mt[r * STRIDE + j]));
                }
            }
        } else if ((mode & 1) && step >= 1 && step <= nchunk && tid < nrows && live[tid]) {
            // ---- stage 2 of chunk `step - 1`: the recurrence, the rows side by side, frames in order
            const int k = step - 1, m0 = k * COLS, nfr = min(COLS, TP - m0);
            __builtin_amdgcn_s_setprio(3);                       // the one dependent chain of the step: issue it ahead of the workers
            const FinCoef c = coef[tid];
            const float* prow = Pb(k % 3) + tid * STRIDE;
            float* mrow = Mb(k & 1) + tid * STRIDE;
            float M = k == 0 ? prow[0] : carry;                  // state starts at p_0 (postprocessing.py:15)
            int j0 = 0;
            for (; j0 + 16 <= nfr; j0 += 16) {                   // full groups: 16 reads in flight, no per-frame guards
                float pv[16];
#pragma unroll
                for (int kk = 0; kk < 16; ++kk) pv[kk] = prow[j0 + kk];
#pragma unroll
                for (int kk = 0; kk < 16; ++kk) {
                    M = fin_ema_step(c, pv[kk], M);
                    mrow[j0 + kk] = M;
                }
            }
            for (; j0 < nfr; ++j0) {
                M = fin_ema_step(c, prow[j0], M);
                mrow[j0] = M;
            }
            carry = M;
            __builtin_amdgcn_s_setprio(0);
        }
        __syncthreads();
    }
}

// the row kernel: 16 rows x 128 frames per chunk by 1024 threads (a 1 s clip's T' = 100 frames is one chunk: three
// dependent stages, each one pass per thread)
// leaf_pcen_stream_f32: one lane per (stream, filter) row, the chunk's frames in order, smoother state in and out
#ifndef LEAF_INST_TU
__global__ void pcen_stream_kernel(const float* __restrict__ p, int BF, int n, FinParams q, const float* __restrict__ ema_in,
                                   float* __restrict__ ema_out) {
    const int row = blockIdx.x * blockDim.x + threadIdx.x;
    if (row >= BF) return;
    const FinCoef c = fin_coef(q, row % q.F);
    const float* pr = p + (size_t)row * n;
    float M = ema_in ? ema_in[row] : pr[0];                     // a stream starts at its first frame (postprocessing.py:15)
    for (int m = 0; m < n; ++m) {
        const float v = pr[m];
        if (q.mode & 1) M = fin_ema_step(c, v, M);
        fin_store(q, (size_t)row * n + m, fin_point(c, q.mode, q.floor_, v, M));
    }
    if (ema_out && (q.mode & 1)) ema_out[row] = M;
}
#endif

// The tail of a workgroup kernel (all its waves call, after their last task): the rows of the clips whose blocks this workgroup
// ran itself, [b_lo F, b_hi F), finalized with `tile_mem` (the waves' scratch, free by now; >= fin_tile_floats<TR, 64>() floats)
// as tile memory.  Release (the partial sums have reached L2) - barrier - tiles.  Rows of up to 128 frames go through in one
// chunk (three dependent stages), longer ones 64 frames at a time, pipelined.
template <int TR>
__device__ __forceinline__ void wg_tail_finalize(const FinParams& fin, int b_lo, int b_hi, float* tile_mem, int tid, int nthreads) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __syncthreads();
    const int row_end = b_hi * fin.F;
    for (int row = b_lo * fin.F; row < row_end; row += TR) {
        if (fin.TP <= 128) fft_finalize_tile<true, TR, 128>(fin, row, min(TR, row_end - row), OwnedClips{}, tile_mem, tid, nthreads);
        else fft_finalize_tile<true, TR, 64>(fin, row, min(TR, row_end - row), OwnedClips{}, tile_mem, tid, nthreads);
    }
}

constexpr int kFinKernelRows = 16, kFinKernelCols = 128;
// NT = 512 when there are many tiles (four workgroups share a CU: throughput), 1024 when there are few (every stage one pass
// per thread: latency -- small batches, long rows)
template <int NT>
__global__ __launch_bounds__(NT) void fft_finalize_kernel(const FinParams q, int B, OwnedClips own) {
    __shared__ float tile[fin_tile_floats<kFinKernelRows, kFinKernelCols>()];
    const int row0 = blockIdx.x * kFinKernelRows, nrows_all = B * q.F;
    if (row0 >= nrows_all) return;
    const int nrows = min(kFinKernelRows, nrows_all - row0);
    if (own.nblocks > 0) {                                       // nothing to do when every clip of the tile was finalized already
        bool any = false;
        for (int b = row0 / q.F; b <= (row0 + nrows - 1) / q.F; ++b) any = any || !own.owned(b);
        if (!any) return;
    }
    fft_finalize_tile<false, kFinKernelRows, kFinKernelCols>(q, row0, nrows, own, tile, threadIdx.x, NT);
}

// (The per-block (d mu, d sigma) partials of the overlap-save backward are summed, in a fixed order, and passed through the
// clamp sub-gradients of convolution.py:15-22 by param_reduce_kernel, leaf_backward.hpp -- a kernel of its own until round 4.)

}  // namespace
