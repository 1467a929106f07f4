// Microbenchmark (GPU box): issue cost of fp32 VALU instructions on gfx950, measured with HIP events.
//   (1) single instructions, 16 independent dependency chains, at 1 / 2 / 3 / 4 waves per SIMD (256..1024-thread workgroups,
//       one per CU: the dynamic LDS request keeps a second workgroup off the CU);
//   (2) the kernel's own 32-point register transform (fft32_dif of leaf_fft.hpp: fma / fmac / fmamk / add / sub with its
//       real dependency distances) in a register-only loop, 1..4 waves per SIMD;
//   (3) "task mix": the whole VALU stream of ONE filter task of leaf_fft_wg_kernel<401,160,12> -- spectral multiply fused
//       with the first DIT stage, 4 stages, 31 twiddle products, the half-wave combine, the fused twiddle + first stage,
//       4 stages, |.|^2, the pooling FMAs with the weights in registers, the frame butterfly -- with every LDS / global
//       operand replaced by a register, 1..4 waves per SIMD (in place: <= 128 VGPRs).  The rate it reaches at
//       the kernel's occupancy (three waves per SIMD) is the PRACTICAL VALU ROOF of that kernel: what is left when every
//       LDS wait, queue spin, table load and s_waitcnt is taken away and only instruction issue remains.
//       Printed as tasks/s for the chip, as executed TFLOP/s (the same flop count per task bench.py uses) and as a fraction
//       of the 157.3 TF issue peak; the last line is JSON for profiles/valu_roof.json.
//   hipcc --offload-arch=gfx950 -O3 -fno-slp-vectorize -I leaf_pytorch_amd/csrc -I include tools/ubench_valu.hip -o /tmp/ubench_valu && /tmp/ubench_valu
#define LEAF_INST_TU 1             // templates and device functions only: none of the library's kernels is compiled here
#include "leaf_fft.hpp"
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x2 __attribute__((ext_vector_type(2)));

template <int MODE>
__global__ __launch_bounds__(1024) void k(float* out, int iters) {
    const int lane = threadIdx.x;
    f32x2 v[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) v[i] = f32x2{(float)(lane + i), (float)(lane - i)};
    const f32x2 m = {1.0001f, 0.9999f}, c = {0.5f, -0.5f};
    const unsigned long long cond = 0x5555555555555555ull + (unsigned long long)iters;
    const float sc = __builtin_amdgcn_readfirstlane(iters) > 0 ? 1.0001f : 0.5f;   // a value the compiler keeps in an SGPR
    (void)sc;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int rep = 0; rep < 4; ++rep)
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                if (MODE == 0) {            // two scalar v_fma_f32
                    asm volatile("v_fma_f32 %0, %0, %2, %3\n v_fma_f32 %1, %1, %2, %3" : "+v"(v[i].x), "+v"(v[i].y) : "v"(m.x), "v"(c.x));
                } else if (MODE == 1) {     // one v_pk_fma_f32
                    asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(v[i]) : "v"(m), "v"(c));
                } else if (MODE == 2) {     // one v_pk_mul_f32
                    asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(v[i]) : "v"(m));
                } else if (MODE == 3) {     // one v_pk_add_f32
                    asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(v[i]) : "v"(c));
                } else if (MODE == 4) {     // one scalar v_fma_f32
                    asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[i].x) : "v"(m.x), "v"(c.x));
                } else if (MODE == 5) {     // one scalar v_add_f32
                    asm volatile("v_add_f32 %0, %0, %1" : "+v"(v[i].x) : "v"(c.x));
                } else if (MODE == 6) {     // one v_permlane32_swap (gfx950)
                    asm volatile("v_permlane32_swap_b32 %0, %1" : "+v"(v[i].x), "+v"(v[i].y));
                } else if (MODE == 7) {     // one v_cndmask_b32 with an SGPR-pair condition (VOP3)
                    asm volatile("v_cndmask_b32_e64 %0, %0, %1, %2" : "+v"(v[i].x) : "v"(c.x), "s"(cond));
                } else if (MODE == 8) {     // v_fmac_f32 (VOP2, 4 bytes): dst += a * b, all VGPRs
                    asm volatile("v_fmac_f32_e32 %0, %1, %2" : "+v"(v[i].x) : "v"(m.x), "v"(v[i].y));
                } else if (MODE == 9) {     // v_fmac_f32 with an SGPR multiplicand (VOP2, 4 bytes)
                    asm volatile("v_fmac_f32_e32 %0, %1, %2" : "+v"(v[i].x) : "s"(sc), "v"(v[i].y));
                } else if (MODE == 10) {    // v_fmamk_f32: dst = a * literal + b (VOP2 + 32-bit literal, 8 bytes)
                    asm volatile("v_fmamk_f32 %0, %0, 0x3f7b14be, %1" : "+v"(v[i].x) : "v"(c.x));
                } else if (MODE == 11) {    // v_mul_f32 (VOP2, 4 bytes)
                    asm volatile("v_mul_f32_e32 %0, %1, %0" : "+v"(v[i].x) : "v"(m.x));
                } else if (MODE == 12) {    // v_fma_f32 with an inline constant and a negated addend: 2 a - p (VOP3, 8 bytes)
                    asm volatile("v_fma_f32 %0, 2.0, %0, -%1" : "+v"(v[i].x) : "v"(v[i].y));
                } else if (MODE == 13) {    // v_sub_f32 (VOP2, 4 bytes)
                    asm volatile("v_sub_f32_e32 %0, %0, %1" : "+v"(v[i].x) : "v"(c.x));
                } else if (MODE == 14) {    // v_fma_f32, three DIFFERENT source registers, separate destination (VOP3, 8 bytes)
                    asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(v[i].x) : "v"(v[i].y), "v"(m.x), "v"(c.x));
                }
            }
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) s += v[i].x + v[i].y;
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

constexpr int kLdsReserve = 96 * 1024;      // more than half a CU's LDS: one workgroup per CU

template <int MODE>
void run(const char* name, int threads, int instr_per_slot) {
    const int iters = 20000, blocks = 256;
    hipFuncSetAttribute(reinterpret_cast<const void*>(k<MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, kLdsReserve);
    float* out; hipMalloc(&out, blocks * 1024 * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float best = 1e9f;
    for (int rep = 0; rep < 4; ++rep) {
        hipEventRecord(e0, 0);
        hipLaunchKernelGGL((k<MODE>), dim3(blocks), dim3(threads), kLdsReserve, 0, out, iters);
        hipEventRecord(e1, 0); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (rep && ms < best) best = ms;
    }
    const double slots = (double)iters * 64 * (threads / 256);       // instruction slots per SIMD (waves/SIMD x per wave)
    const double ns_per = best * 1e6 / (slots * instr_per_slot);
    printf("%-22s waves/SIMD=%d  %.3f ms  -> %.3f ns per instruction per SIMD  (= %.2f cycles at 2.4 GHz)\n", name,
           threads / 256, best, ns_per, ns_per * 2.4);
    hipFree(out);
}

// ---- (2) the 32-point register transform, register-only ---------------------------------------------------------------
template <int WAVES>
__global__ __launch_bounds__(WAVES * 256, 1) void k_fft32(float* out, int iters) {
    const int lane = threadIdx.x & 63;
    float re[32], im[32];
#pragma unroll
    for (int i = 0; i < 32; ++i) { re[i] = 1e-3f * (float)(lane + i); im[i] = 1e-3f * (float)(lane - i); }
    for (int it = 0; it < iters; ++it) {
        fft32_dif(re, im);
        pin32(re);
        pin32(im);
#pragma unroll
        for (int i = 0; i < 32; ++i) { re[i] *= 0.03125f; }          // keep magnitudes bounded (32 v_mul: counted below)
        pin32(re);
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 32; ++i) s += re[i] + im[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

// ---- (2b) the same 32-point transform on COMPLEX PAIRS with packed fp32 (v_pk_add_f32 / v_pk_fma_f32, op_sel swaps and neg
// modifiers instead of shuffles): 194 packed instructions instead of 388 scalar ones.  Not what the kernels run -- a measurement
// for the next design step: a packed instruction takes one issue slot for two flops per lane, and at three waves per SIMD the
// issue slots, not the ALU, are what is short.
typedef float c32 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void pk_bfly_trivial(c32& a, c32& b) {          // w = 1: a' = a + b, b' = a - b
    c32 s;
    asm volatile("v_pk_add_f32 %0, %1, %2" : "=v"(s) : "v"(a), "v"(b));
    asm volatile("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(b) : "v"(a), "v"(b));
    a = s;
}
__device__ __forceinline__ void pk_bfly_mi(c32& a, c32& b) {               // w = -i: w b = (bi, -br)
    c32 s;
    asm volatile("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_hi:[0,1]" : "=v"(s) : "v"(a), "v"(b));
    asm volatile("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1]" : "=v"(b) : "v"(a), "v"(b));
    a = s;
}
__device__ __forceinline__ void pk_bfly(c32& a, c32& b, const c32& w, const c32& two) {   // general w = (c, s) in a register pair
    c32 t;
    asm volatile("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,0,0] op_sel_hi:[1,0,1]" : "=v"(t) : "v"(b), "v"(w), "v"(a));             // a + b (c, c)
    asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,1,0] op_sel_hi:[0,1,1] neg_lo:[0,1,0]" : "+v"(t) : "v"(b), "v"(w));    // + (bi, br) (-s, s)
    asm volatile("v_pk_fma_f32 %0, %1, %2, %3 neg_lo:[0,0,1] neg_hi:[0,0,1]" : "=v"(b) : "v"(two), "v"(a), "v"(t));            // 2 a - a'
    a = t;
}
template <int WAVES>
__global__ __launch_bounds__(WAVES * 256, 1) void k_fft32pk(float* out, int iters, float seed) {
    const int lane = threadIdx.x & 63;
    c32 z[32], w[8];
#pragma unroll
    for (int i = 0; i < 32; ++i) z[i] = c32{1e-3f * (float)(lane + i), 1e-3f * (float)(lane - i)};
#pragma unroll
    for (int i = 0; i < 8; ++i) w[i] = c32{__cosf(seed * (float)(i + 1)), -__sinf(seed * (float)(i + 1))};
    const c32 two = {2.0f, 2.0f}, sc = {0.03125f, 0.03125f};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int half = 1; half <= 16; half <<= 1)
#pragma unroll
            for (int blk = 0; blk < 32; blk += 2 * half)
#pragma unroll
                for (int j = 0; j < half; ++j) {
                    const int tw = j * (16 / half);
                    if (tw == 0) pk_bfly_trivial(z[blk + j], z[blk + j + half]);
                    else if (tw == 8) pk_bfly_mi(z[blk + j], z[blk + j + half]);
                    else pk_bfly(z[blk + j], z[blk + j + half], w[tw & 7], two);
                }
#pragma unroll
        for (int i = 0; i < 32; i += 2) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(z[i]) : "v"(sc));   // rescale (16 packed: counted)
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 32; ++i) s += z[i].x + z[i].y;
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

// ---- (3) one filter task of leaf_fft_wg_kernel<401,160,12>, VALU stream only ---------------------------------------------
// Geometry constants of the 16 kHz LEAF window on 2048-sample blocks (leaf_fft_wg.hpp): L = 1600, 25 rows of |y|^2, 13 frames,
// 15 distinct pooling-weight vectors; each (row, frame) pair whose window meets the row is one FMA (80 of them).
template <int WAVES>
__global__ __launch_bounds__(WAVES * 256, 1) void k_task(float* out, int iters, float seed) {
    constexpr int SK = 401, SHOP = 160, PADL = 200, LS = 1600, NROW = 25;
    constexpr int DMIN = -((SK - 1 - PADL) / SHOP), DMAX = (LS - 1 + PADL) / SHOP, NFR = DMAX - DMIN + 1;
    constexpr int PG = 32, PJ0 = -55, NJ = 15;                           // wg_pool_step / _jmin / _nj at 401 / 160
    int lane = threadIdx.x & 63;
    asm volatile("" : "+v"(lane));
    const int h = lane >> 5;
    const float sg = h ? -1.0f : 1.0f;
    float zre[32], zim[32], rq[32], pw[NJ];
    float wx = __cosf(seed * (float)lane), wy = __sinf(seed * (float)lane);     // a unit twiddle per lane (loop-invariant)
#pragma unroll
    for (int i = 0; i < 32; ++i) { zre[i] = 1e-3f * (float)(lane + i); zim[i] = 1e-3f * (float)(lane - i); rq[i] = 9.5e-4f + 1e-6f * (float)i; }
#pragma unroll
    for (int k2 = 0; k2 < NJ; ++k2) pw[k2] = 1e-3f * (float)(k2 + 1);
    pin32(rq);                                                           // table values live in registers, not as literals
#pragma unroll
    for (int k2 = 0; k2 < NJ; ++k2) asm volatile("" : "+v"(pw[k2]));
    float carry = 0.0f;
    for (int it = 0; it < iters; ++it) {
        // spectral multiply fused with the first DIT stage: 16 pairs x 6 instructions (operands: registers instead of the ring;
        // in place, so that the whole loop fits the 128 VGPRs of four waves per SIMD)
#pragma unroll
        for (int k2 = 0; k2 < 16; ++k2) {
            const float ra = rq[k2], rb = rq[k2 + 16];
            const float lr = zre[k2], li = zim[k2], hr = zre[k2 + 16], hi_ = zim[k2 + 16];
            const float tr_ = lr * ra, ti_ = -(li * ra);
            zre[k2] = fmaf(hr, rb, tr_);
            zim[k2] = fmaf(hi_, rb, ti_);
            zre[k2 + 16] = fmaf(-hr, rb, tr_);
            zim[k2 + 16] = fmaf(-hi_, rb, ti_);
        }
        pin32(zre); pin32(zim);
        fft32_dit_stage<2>(zre, zim); fft32_dit_stage<4>(zre, zim); fft32_dit_stage<8>(zre, zim); fft32_dit_stage<16>(zre, zim);
        // 31 first-level twiddle products (4 instructions each)
#pragma unroll
        for (int i = 1; i < 32; ++i) {
            const float r = zre[i] * wx - zim[i] * wy;
            zim[i] = zre[i] * wy + zim[i] * wx;
            zre[i] = r;
        }
        pin32(zre); pin32(zim);
        // half-wave combine t = a + sg b (64 FMAs; a, b: the two transposed reads -- here the register itself and a constant)
        float (&tr)[32] = zre;
        float (&ti)[32] = zim;
#pragma unroll
        for (int i = 0; i < 32; ++i) { tr[i] = fmaf(wy, sg, zre[i]); ti[i] = fmaf(wx, sg, zim[i]); }
        pin32(tr); pin32(ti);
        // half-wave twiddle fused into the first DIT stage of the second transform: 16 pairs x 10 instructions
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            float ar, ai;
            if (j == 0) { ar = tr[0]; ai = ti[0]; }
            else { ar = tr[j] * wx - ti[j] * wy; ai = tr[j] * wy + ti[j] * wx; }
            const float br = tr[j + 16], bi = ti[j + 16];
            const float pr = fmaf(-bi, wy, fmaf(br, wx, ar));
            const float pi = fmaf(bi, wx, fmaf(br, wy, ai));
            zre[j] = pr; zim[j] = pi;                                     // (tr / ti alias zre / zim: in place, like the kernel's register reuse)
            zre[j + 16] = fmaf(2.0f, ar, -pr); zim[j + 16] = fmaf(2.0f, ai, -pi);
        }
        fft32_dit_stage<2>(zre, zim); fft32_dit_stage<4>(zre, zim); fft32_dit_stage<8>(zre, zim); fft32_dit_stage<16>(zre, zim);
        pin32(zre); pin32(zim);
        // |y|^2 of the 25 valid rows, the 80 pooling FMAs with the weights in registers, the frame butterfly
        float er[NROW];
#pragma unroll
        for (int i = 0; i < 32; ++i) {
            const int r = brev5(i);
            if (r < NROW) er[r] = zre[i] * zre[i] + zim[i] * zim[i];
        }
        float acc[16];
#pragma unroll
        for (int fi = 0; fi < 16; ++fi) acc[fi] = 0.0f;
#pragma unroll
        for (int r = 0; r < NROW; ++r)
#pragma unroll
            for (int fi = 0; fi < NFR; ++fi) {
                const int is = (DMIN + fi) * SHOP - PADL;
                if (is <= 64 * r + 63 && is + SK > 64 * r) acc[fi] = fmaf(er[r], pw[(64 * r - is - PJ0) / PG], acc[fi]);
            }
        asm volatile("" : "+v"(acc[0]));
        carry += frame_butterfly16(acc, lane);
        // feed back: keeps every value live and bounded (two transforms gain <= 1024, rq ~ 1e-3)
        zre[0] = fmaf(carry, 1e-30f, zre[0]);
        pin32(zre); pin32(zim);
    }
    float s = carry;
#pragma unroll
    for (int i = 0; i < 32; ++i) s += zre[i] + zim[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <typename Kern>
float time_kernel(Kern kern, int threads, int iters, float* out, float extra) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float best = 1e9f;
    for (int rep = 0; rep < 4; ++rep) {
        hipEventRecord(e0, 0);
        kern(threads, iters, out, extra);
        hipEventRecord(e1, 0); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (rep && ms < best) best = ms;
    }
    return best;
}

template <int W>
void run_fft32(float* out) {
    const int iters = 4000;
    hipFuncSetAttribute(reinterpret_cast<const void*>(k_fft32<W>), hipFuncAttributeMaxDynamicSharedMemorySize, kLdsReserve);
    const float ms = time_kernel([](int th, int it, float* o, float) { hipLaunchKernelGGL((k_fft32<W>), dim3(256), dim3(th), kLdsReserve, 0, o, it); },
                                 W * 256, iters, out, 0.f);
    const double instr = 388.0 + 32.0;                                    // per transform: the butterflies + the rescale
    const double ns_per = ms * 1e6 / ((double)iters * W * instr);
    printf("fft32_dif (388+32 VALU)  waves/SIMD=%d  %.3f ms  -> %.3f ns per instruction per SIMD  (= %.2f cycles at 2.4 GHz; %.0f SIMD cycles per transform)\n",
           W, ms, ns_per, ns_per * 2.4, ns_per * 2.4 * instr);
}

template <int W>
void run_fft32pk(float* out) {
    const int iters = 4000;
    hipFuncSetAttribute(reinterpret_cast<const void*>(k_fft32pk<W>), hipFuncAttributeMaxDynamicSharedMemorySize, kLdsReserve);
    const float ms = time_kernel([](int th, int it, float* o, float sd) { hipLaunchKernelGGL((k_fft32pk<W>), dim3(256), dim3(th), kLdsReserve, 0, o, it, sd); },
                                 W * 256, iters, out, 0.37f);
    const double per_transform_cycles = ms * 1e6 / ((double)iters * W) * 2.4;     // SIMD cycles one transform occupies
    printf("fft32 on complex pairs, packed (194+16 v_pk_*)  waves/SIMD=%d  %.3f ms  -> %.0f SIMD cycles per transform (scalar form: see fft32_dif x 420)\n",
           W, ms, per_transform_cycles);
}

// executed flops of one (block, filter) task as bench.py counts them: inverse transform + multiply + |.|^2 + pooling
constexpr double kTaskFlops = 5.0 * 2048 * 11 + 5.0 * 2048 + 2.0 * 64 * 8 * 14;
double g_task_rate[5];
template <int W>
void run_task(float* out) {
    const int iters = 2000;
    hipFuncSetAttribute(reinterpret_cast<const void*>(k_task<W>), hipFuncAttributeMaxDynamicSharedMemorySize, kLdsReserve);
    const float ms = time_kernel([](int th, int it, float* o, float sd) { hipLaunchKernelGGL((k_task<W>), dim3(256), dim3(th), kLdsReserve, 0, o, it, sd); },
                                 W * 256, iters, out, 0.37f);
    const double tasks_per_s = 256.0 * 4 * W * iters / (ms * 1e-3);       // whole chip: 1024 SIMDs x W waves
    const double tf = tasks_per_s * kTaskFlops / 1e12;
    g_task_rate[W] = tf / 157.3;
    printf("filter-task VALU mix     waves/SIMD=%d  %.3f ms  -> %.2f us per task per wave, %.1f M tasks/s, %.1f TFLOP/s executed = %.3f of 157.3 TF\n",
           W, ms, ms * 1e3 / iters, tasks_per_s / 1e6, tf, tf / 157.3);
}

int main() {
    for (int t : {256, 512, 768, 1024}) {
        run<0>("2x v_fma_f32", t, 2);
        run<4>("v_fma_f32", t, 1);
        run<5>("v_add_f32", t, 1);
        run<1>("v_pk_fma_f32", t, 1);
        run<2>("v_pk_mul_f32", t, 1);
        run<3>("v_pk_add_f32", t, 1);
        run<6>("v_permlane32_swap_b32", t, 1);
        run<7>("v_cndmask_b32_e64", t, 1);
        run<8>("v_fmac_f32 (VOP2 4B)", t, 1);
        run<9>("v_fmac_f32 sgpr src0", t, 1);
        run<10>("v_fmamk_f32 (8B)", t, 1);
        run<11>("v_mul_f32 (4B)", t, 1);
        run<12>("v_fma 2.0*a-p (8B)", t, 1);
        run<13>("v_sub_f32 (4B)", t, 1);
        run<14>("v_fma 3 srcs (8B)", t, 1);
    }
    float* out; hipMalloc(&out, 256 * 1024 * 4);
    run_fft32<1>(out); run_fft32<2>(out); run_fft32<3>(out); run_fft32<4>(out);
    run_fft32pk<1>(out); run_fft32pk<2>(out); run_fft32pk<3>(out); run_fft32pk<4>(out);
    run_task<1>(out); run_task<2>(out); run_task<3>(out); run_task<4>(out);
    printf("{\"kernel\": \"leaf_fft_wg_kernel<401,160,12>\", \"waves_per_simd\": 3, \"frac_of_peak\": %.4f, "
           "\"frac_of_peak_by_waves\": {\"1\": %.4f, \"2\": %.4f, \"3\": %.4f, \"4\": %.4f}, \"peak_TFLOPs\": 157.3, "
           "\"what\": \"register-only restatement of one filter task's VALU stream (tools/ubench_valu.hip), executed flops per task as bench.py counts them\"}\n",
           g_task_rate[3], g_task_rate[1], g_task_rate[2], g_task_rate[3], g_task_rate[4]);
    return 0;
}
