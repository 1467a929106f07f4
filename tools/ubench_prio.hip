// Microbenchmark (GPU box), round 6: the three-waves-per-SIMD issue penalty and wave priorities.
// tools/ubench_valu.hip measured the kernel's 32-point register transform at 908 / 1182 / 883 SIMD cycles per transform with 2 / 3 / 4
// waves per SIMD: with THREE equal waves a SIMD issues ~30 % slower than with two or four.  leaf_fft_wg_kernel runs three (twelve-wave
// workgroups: LDS latency hiding measured better than eight waves).  Question: does s_setprio turn a three-wave SIMD into "two waves at
// the two-wave cadence + a third that fills their stalls"?
// Streams: (a) fft32_dif, register-only; (b) the filter-task VALU mix of ubench_valu.hip; (c) the same mix with the kernel's two LDS
// transpositions per task (32 ds_write_b32 + wait + 8 ds_read_b128 + wait, wave-private scratch) so that waves really stall.
// Priority schemes by the wave's slot on its SIMD (slot = wave / 4: waves w, w + 4, w + 8 share SIMD w):
//     equal      0 0 0          ladder   2 1 0         two-high   1 1 0        one-high   1 0 0
//   hipcc --offload-arch=gfx950 -O3 -fno-slp-vectorize -I leaf_pytorch_amd/csrc -I include tools/ubench_prio.hip -o /tmp/ubench_prio && /tmp/ubench_prio
#define LEAF_INST_TU 1
#include "leaf_fft.hpp"
#include <hip/hip_runtime.h>
#include <cstdio>

template <int SCHEME>
__device__ __forceinline__ void set_prio(int slot) {
    if (SCHEME == 1) { if (slot == 0) __builtin_amdgcn_s_setprio(2); else if (slot == 1) __builtin_amdgcn_s_setprio(1); }
    if (SCHEME == 2) { if (slot < 2) __builtin_amdgcn_s_setprio(1); }
    if (SCHEME == 3) { if (slot == 0) __builtin_amdgcn_s_setprio(1); }
}

template <int WAVES, int SCHEME>
__global__ __launch_bounds__(WAVES * 256, 1) void k_fft32(float* out, int iters, unsigned long long* cyc) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    set_prio<SCHEME>(wave >> 2);
    float re[32], im[32];
#pragma unroll
    for (int i = 0; i < 32; ++i) { re[i] = 1e-3f * (float)(lane + i); im[i] = 1e-3f * (float)(lane - i); }
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
        fft32_dif(re, im);
        pin32(re);
        pin32(im);
#pragma unroll
        for (int i = 0; i < 32; ++i) { re[i] *= 0.03125f; }
        pin32(re);
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 32; ++i) s += re[i] + im[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (lane == 0 && blockIdx.x == 0) cyc[wave] = t1 - t0;
}

// one filter task of leaf_fft_wg_kernel<401,160,12> (ubench_valu.hip's k_task), optionally with two LDS transpositions
template <int WAVES, int SCHEME, bool LDS>
__global__ __launch_bounds__(WAVES * 256, 1) void k_task(float* out, int iters, float seed, unsigned long long* cyc) {
    extern __shared__ __attribute__((aligned(16))) float dyn[];
    constexpr int SK = 401, SHOP = 160, PADL = 200, LS = 1600, NROW = 25;
    constexpr int DMIN = -((SK - 1 - PADL) / SHOP), DMAX = (LS - 1 + PADL) / SHOP, NFR = DMAX - DMIN + 1;
    constexpr int PG = 32, PJ0 = -55, NJ = 15;
    int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    asm volatile("" : "+v"(lane));
    set_prio<SCHEME>(wave >> 2);
    float* scr = dyn + (size_t)wave * (32 * 68);                          // the kernel's [32][68] wave-private scratch
    const int h = lane >> 5;
    const float sg = h ? -1.0f : 1.0f;
    float zre[32], zim[32], rq[32], pw[NJ];
    float wx = __cosf(seed * (float)lane), wy = __sinf(seed * (float)lane);
#pragma unroll
    for (int i = 0; i < 32; ++i) { zre[i] = 1e-3f * (float)(lane + i); zim[i] = 1e-3f * (float)(lane - i); rq[i] = 9.5e-4f + 1e-6f * (float)i; }
#pragma unroll
    for (int k2 = 0; k2 < NJ; ++k2) pw[k2] = 1e-3f * (float)(k2 + 1);
    pin32(rq);
#pragma unroll
    for (int k2 = 0; k2 < NJ; ++k2) asm volatile("" : "+v"(pw[k2]));
    float carry = 0.0f;
    // a transposition as the kernel does it per plane: 32 stores (row i, column lane), wait, the lane's row back as 8 x 16 bytes, wait
    auto transpose = [&](float (&v)[32]) {
        if (!LDS) return;
#pragma unroll
        for (int i = 0; i < 32; ++i) scr[i * 68 + lane] = v[i];
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        const f32x4* row = reinterpret_cast<const f32x4*>(scr + (lane & 31) * 68 + 32 * (lane >> 5));
#pragma unroll
        for (int q = 0; q < 8; ++q) { const f32x4 t = row[q]; v[4 * q] = t.x; v[4 * q + 1] = t.y; v[4 * q + 2] = t.z; v[4 * q + 3] = t.w; }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    };
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
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
#pragma unroll
        for (int i = 1; i < 32; ++i) {
            const float r = zre[i] * wx - zim[i] * wy;
            zim[i] = zre[i] * wy + zim[i] * wx;
            zre[i] = r;
        }
        pin32(zre); pin32(zim);
        transpose(zre);
        transpose(zim);
        float (&tr)[32] = zre;
        float (&ti)[32] = zim;
#pragma unroll
        for (int i = 0; i < 32; ++i) { tr[i] = fmaf(wy, sg, zre[i]); ti[i] = fmaf(wx, sg, zim[i]); }
        pin32(tr); pin32(ti);
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            float ar, ai;
            if (j == 0) { ar = tr[0]; ai = ti[0]; }
            else { ar = tr[j] * wx - ti[j] * wy; ai = tr[j] * wy + ti[j] * wx; }
            const float br = tr[j + 16], bi = ti[j + 16];
            const float pr = fmaf(-bi, wy, fmaf(br, wx, ar));
            const float pi = fmaf(bi, wx, fmaf(br, wy, ai));
            zre[j] = pr; zim[j] = pi;
            zre[j + 16] = fmaf(2.0f, ar, -pr); zim[j + 16] = fmaf(2.0f, ai, -pi);
        }
        fft32_dit_stage<2>(zre, zim); fft32_dit_stage<4>(zre, zim); fft32_dit_stage<8>(zre, zim); fft32_dit_stage<16>(zre, zim);
        pin32(zre); pin32(zim);
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
        zre[0] = fmaf(carry, 1e-30f, zre[0]);
        pin32(zre); pin32(zim);
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = carry;
#pragma unroll
    for (int i = 0; i < 32; ++i) s += zre[i] + zim[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (lane == 0 && blockIdx.x == 0) cyc[wave] = t1 - t0;
}

constexpr int kLds = 16 * 32 * 68 * 4;      // scratch of up to 16 waves (139 KB): one workgroup per CU either way

template <typename L>
float time_it(L launch) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float best = 1e9f;
    for (int rep = 0; rep < 4; ++rep) {
        hipEventRecord(e0, 0);
        launch();
        hipEventRecord(e1, 0); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (rep && ms < best) best = ms;
    }
    return best;
}

float* g_out;
unsigned long long* g_cyc;
const char* kScheme[4] = {"equal    0 0 0", "ladder   2 1 0", "two-high 1 1 0", "one-high 1 0 0"};

void per_slot(int waves, char* buf) {
    unsigned long long h[16];
    hipMemcpy(h, g_cyc, sizeof(h), hipMemcpyDeviceToHost);
    int n = 0;
    for (int slot = 0; slot < waves; ++slot) {
        double s = 0;
        for (int w = 0; w < 4; ++w) s += (double)h[4 * slot + w];
        n += sprintf(buf + n, "%s%.2f", slot ? " / " : "", s / 4 / (double)h[0]);
    }
}

template <int W, int S>
void run_fft() {
    const int iters = 4000;
    hipFuncSetAttribute(reinterpret_cast<const void*>(k_fft32<W, S>), hipFuncAttributeMaxDynamicSharedMemorySize, kLds);
    const float ms = time_it([&] { hipLaunchKernelGGL((k_fft32<W, S>), dim3(256), dim3(W * 256), kLds, 0, g_out, iters, g_cyc); });
    char buf[128];
    per_slot(W, buf);
    printf("fft32_dif        waves/SIMD=%d  %-15s %7.3f ms -> %5.0f SIMD cycles per transform at 2.4 GHz   (time in the loop by slot, relative: %s)\n", W,
           kScheme[S], ms, ms * 1e6 / ((double)iters * W) * 2.4, buf);
}
template <int W, int S, bool LDS>
double run_task() {
    const int iters = 2000;
    hipFuncSetAttribute(reinterpret_cast<const void*>(k_task<W, S, LDS>), hipFuncAttributeMaxDynamicSharedMemorySize, kLds);
    const float ms = time_it([&] { hipLaunchKernelGGL((k_task<W, S, LDS>), dim3(256), dim3(W * 256), kLds, 0, g_out, iters, 0.37f, g_cyc); });
    char buf[128];
    per_slot(W, buf);
    const double us_per_task_per_simd = ms * 1e3 / ((double)iters * W);
    printf("filter task %-4s waves/SIMD=%d  %-15s %7.3f ms -> %.3f us per task per SIMD   (by slot: %s)\n", LDS ? "+LDS" : "", W, kScheme[S], ms,
           us_per_task_per_simd, buf);
    return us_per_task_per_simd;
}

int main() {
    hipMalloc(&g_out, 256 * 1024 * 4);
    hipMalloc(&g_cyc, 16 * 8);
    run_fft<2, 0>(); run_fft<4, 0>();
    run_fft<3, 0>(); run_fft<3, 1>(); run_fft<3, 2>(); run_fft<3, 3>();
    run_task<2, 0, false>(); run_task<4, 0, false>();
    const double a = run_task<3, 0, false>(), b = run_task<3, 1, false>(), c = run_task<3, 2, false>(), d = run_task<3, 3, false>();
    run_task<2, 0, true>(); run_task<4, 0, true>();
    const double e = run_task<3, 0, true>(), f = run_task<3, 1, true>(), g = run_task<3, 2, true>(), h = run_task<3, 3, true>();
    run_task<4, 1, true>();
    printf("{\"task_us_per_simd_3waves\": {\"equal\": %.4f, \"ladder\": %.4f, \"two_high\": %.4f, \"one_high\": %.4f}, "
           "\"with_lds\": {\"equal\": %.4f, \"ladder\": %.4f, \"two_high\": %.4f, \"one_high\": %.4f}}\n", a, b, c, d, e, f, g, h);
    return 0;
}
