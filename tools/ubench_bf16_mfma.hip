// Microbenchmark (GPU box), VERDICT r4 next #6: would a bf16-MFMA direct form serve BASELINE configs[4] ("bf16 forward") better
// than the fp32 overlap-save path?  The inner loop of the Hermitian-halved Toeplitz formulation of convolution.py:71-99 on
// v_mfma_f32_32x32x16_bf16, INCLUDING the operand preparation the formulation needs per MFMA:
//     Re y[f][n] = sum_k hr[f][k] s_k[n],  Im y[f][n] = sum_k hi[f][k] d_k[n],  s / d = x[n + k] +- x[n - k],  k = 0 .. 200 (K = 401)
// per (16-tap, 32-sample) tile: 4 ds_read_b128 of the waveform window (fp32 in LDS), 8 adds + 8 subtractions, 8 v_cvt_pk_bf16_f32
// (the B operands s and d), 4 ds_read_b128 of the tap tiles (A operands, shared by NT sample tiles) and 4 MFMAs (two 32-filter
// tiles -- 40 filters padded to 64 -- x {Re, Im}).  Prints shader cycles per MFMA and the MFMA pipe utilisation (32 cycles per
// 32x32x16 bf16 MFMA per SIMD: MI355X_MICROARCH.md) for 1 and 2 waves per SIMD and NT = 1, 2 sample tiles per wave, and the
// whole-forward time that rate projects to for 256 x 10 s clips.
//   hipcc --offload-arch=gfx950 -O3 tools/ubench_bf16_mfma.hip -o /tmp/ubench_bf16_mfma && /tmp/ubench_bf16_mfma
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef short s16x8 __attribute__((ext_vector_type(8)));

constexpr int KT = 13;                 // 16-tap tiles: 201 Hermitian taps padded to 208
constexpr int XW = 4096;               // waveform window in LDS (floats)

__device__ __forceinline__ unsigned pk_bf16(float a, float b) {
    unsigned r;
    asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}

template <int NT>
__global__ __launch_bounds__(512) void k(float* out, unsigned long long* cyc, int iters) {
    __shared__ __attribute__((aligned(16))) float xs[XW];
    __shared__ __attribute__((aligned(16))) unsigned taps[KT][4][64][4];        // [tap tile][2 filter tiles x {re, im}][lane][8 bf16]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < XW; i += blockDim.x) xs[i] = 1e-3f * (float)((i * 7919) & 1023) - 0.5f;
    for (int i = tid; i < KT * 4 * 64 * 4; i += blockDim.x) (&taps[0][0][0][0])[i] = 0x3c003c00u + (unsigned)i;
    __syncthreads();
    f32x16 acc[NT][4];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[t][m][e] = 0.0f;
    const int col = lane & 31, kg = lane >> 5;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
        const int n0 = 1024 + 32 * NT * ((wave + it) & 15);                     // this wave's sample tiles
#pragma unroll 1
        for (int kt = 0; kt < KT; ++kt) {
            const f32x4* ta = reinterpret_cast<const f32x4*>(&taps[kt][0][lane][0]);
            bf16x8 a[4];
#pragma unroll
            for (int m = 0; m < 4; ++m) {
                const f32x4 v = ta[64 * m];
                a[m] = __builtin_bit_cast(bf16x8, v);
            }
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                const int c = n0 + 32 * t + col, k0 = 16 * kt + 8 * kg;
                const f32x4* xp = reinterpret_cast<const f32x4*>(&xs[(c + k0) & ~3]);          // x[n + k .. n + k + 7]
                const f32x4* xm = reinterpret_cast<const f32x4*>(&xs[(c - k0 - 7) & ~3]);      // x[n - k - 7 .. n - k]
                const f32x4 p0 = xp[0], p1 = xp[1], m0 = xm[0], m1 = xm[1];
                const float pv[8] = {p0.x, p0.y, p0.z, p0.w, p1.x, p1.y, p1.z, p1.w};
                const float mv[8] = {m1.w, m1.z, m1.y, m1.x, m0.w, m0.z, m0.y, m0.x};
                unsigned s[4], d[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    s[j] = pk_bf16(pv[2 * j] + mv[2 * j], pv[2 * j + 1] + mv[2 * j + 1]);
                    d[j] = pk_bf16(pv[2 * j] - mv[2 * j], pv[2 * j + 1] - mv[2 * j + 1]);
                }
                const bf16x8 bs = __builtin_bit_cast(bf16x8, f32x4{__uint_as_float(s[0]), __uint_as_float(s[1]), __uint_as_float(s[2]), __uint_as_float(s[3])});
                const bf16x8 bd = __builtin_bit_cast(bf16x8, f32x4{__uint_as_float(d[0]), __uint_as_float(d[1]), __uint_as_float(d[2]), __uint_as_float(d[3])});
                acc[t][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], bs, acc[t][0], 0, 0, 0);     // Re, filters 0..31
                acc[t][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], bd, acc[t][1], 0, 0, 0);     // Im
                acc[t][2] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[2], bs, acc[t][2], 0, 0, 0);     // Re, filters 32..63
                acc[t][3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[3], bd, acc[t][3], 0, 0, 0);     // Im
            }
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float sum = 0.0f;
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
            for (int e = 0; e < 16; ++e) sum += acc[t][m][e];
    out[blockIdx.x * blockDim.x + tid] = sum;
    if (lane == 0) cyc[blockIdx.x * (blockDim.x >> 6) + wave] = t1 - t0;
}

template <int NT>
double run(int threads, int iters) {
    float* out;
    unsigned long long* cyc;
    const int blocks = 256;
    (void)hipMalloc(&out, sizeof(float) * blocks * threads);
    (void)hipMalloc(&cyc, sizeof(unsigned long long) * blocks * (threads / 64));
    hipLaunchKernelGGL(k<NT>, dim3(blocks), dim3(threads), 0, 0, out, cyc, 4);
    hipLaunchKernelGGL(k<NT>, dim3(blocks), dim3(threads), 0, 0, out, cyc, iters);
    (void)hipDeviceSynchronize();
    std::vector<unsigned long long> h(blocks * (threads / 64));
    (void)hipMemcpy(h.data(), cyc, h.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost);
    double s = 0;
    for (auto v : h) s += (double)v;
    (void)hipFree(out);
    (void)hipFree(cyc);
    return s / h.size() / ((double)iters * KT * NT * 4);            // shader cycles per MFMA of one wave
}

int main() {
    printf("v_mfma_f32_32x32x16_bf16 with the Hermitian-Toeplitz operand preparation (s / d = x[n+k] +- x[n-k] from an fp32 LDS window -> bf16),\n"
           "4 MFMAs per (16-tap, 32-sample) tile: shader cycles per MFMA per wave, and the MFMA pipe share of its SIMD (32 cycles per MFMA):\n");
    double best = 0;
    for (int wps = 1; wps <= 2; ++wps) {
        const double c1 = run<1>(256 * wps, 400), c2 = run<2>(256 * wps, 400);
        const double u1 = 32.0 * wps / c1, u2 = 32.0 * wps / c2;
        printf("  %d wave(s) per SIMD: 1 sample tile per wave %.1f cycles / MFMA (pipe %.2f);  2 sample tiles (tap tiles shared) %.1f (pipe %.2f)\n", wps, c1, u1, c2, u2);
        best = u1 > best ? u1 : best;
        best = u2 > best ? u2 : best;
    }
    // BASELINE configs[4] per GPU: 256 clips x 160000 samples, 40 filters padded to 64, 208 Hermitian taps x {Re, Im}
    const double mfmas = 256.0 * 160000 / 32 * KT * 4, cycles = mfmas * 32 / 1024;        // per SIMD (1024 SIMDs)
    printf("configs[4] (256 x 10 s): %.3g MFMAs = %.3f ms of the MFMA pipes at 2.4 GHz; at the best measured pipe share %.2f: %.3f ms for the k-loop alone\n"
           "(no |y|^2, pooling, PCEN, no waveform staging, no tail effects); the fp32 overlap-save path with band tasks runs the WHOLE forward in ~1.06 ms\n",
           mfmas, cycles / 2.4e9 * 1e3, best, cycles / 2.4e9 * 1e3 / best);
    printf("{\"best_mfma_pipe_share\": %.3f, \"kloop_ms_at_2.4GHz\": %.3f}\n", best, cycles / 2.4e9 * 1e3 / best);
    return 0;
}
