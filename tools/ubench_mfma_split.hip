// Microbenchmark (GPU box), VERDICT r5 next #3: would wave-specialised fp16-split MFMA Toeplitz tasks for the wide-band Gabor filters
// (the 13 of 40 default filters the band classes leave on 2048-point transforms) fit BESIDE VALU-only FFT waves on the same CU?
//
// The formulation under test (convolution.py:71-99 as a GEMM, no Hermitian s / d step, no VALU operand preparation):
//     y[r][n] = sum_k h[r][k] x[n + k],   r = 26 rows (13 filters x {Re, Im}) padded to 32,  k = sigma-truncated taps
//   x split ONCE per block into fp16 hi + 2^11 lo in LDS, h split once per call: three v_mfma_f32_32x32x16_f16 per (32-sample,
//   16-tap) tile (hi hi, hi lo, lo hi).  The B operand of lane (n, kg) is x[n0 + n + k0 + 8 kg .. + 7]: eight consecutive fp16 at an
//   address that is only 2-byte aligned (Toeplitz), i.e. ONE ds_read_b128 at an unaligned address per operand -- or eight shifted
//   copies of the window in LDS (8 x 4.9 KB x {hi, lo}) so that every read is 16-byte aligned.  Both are measured.
//
// Kernel: one 12-wave workgroup per CU (the main kernel's shape).  Waves 0 .. NM-1 (one or two per SIMD) run the MFMA k-loop
// (A tiles in registers, two ds_read_b128 + three MFMAs per k-tile), the other waves run the register-only restatement of the
// filter-task VALU stream used by tools/ubench_valu.hip (dependent FMA chains, 32 live accumulators).  Reported: shader cycles per
// MFMA with and without the VALU waves, VALU instructions per cycle per SIMD with and without the MFMA waves, and what the
// measured MFMA rate means for one 1600-sample block: 50 sample tiles x KT tap tiles x 3 MFMAs.
//   hipcc --offload-arch=gfx950 -O3 tools/ubench_mfma_split.hip -o /tmp/ubench_mfma_split && /tmp/ubench_mfma_split
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

constexpr int XW = 2560;               // fp16 window of one block (2048 + K - 1 rounded up), per copy
constexpr int KT = 18;                 // 16-tap tiles: the widest default filter (sigma ~ 25) needs 2 x 5.68 sigma + 1 = 285 taps

// MODE 0: unaligned Toeplitz reads from ONE copy of the window; 1: eight shifted copies, aligned reads
template <int MODE, int NM, bool VALU_ON, bool MFMA_ON>
__global__ __launch_bounds__(768) void k(float* out, unsigned long long* cyc, int iters) {
    __shared__ __attribute__((aligned(16))) _Float16 xh[(MODE ? 8 : 1) * XW], xl[(MODE ? 8 : 1) * XW];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < (MODE ? 8 : 1) * XW; i += blockDim.x) {
        xh[i] = (_Float16)(1e-3f * (float)((i * 7919) & 1023) - 0.5f);
        xl[i] = (_Float16)(1e-3f * (float)((i * 104729) & 1023) - 0.5f);
    }
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    float sum = 0.0f;
    if (wave < NM) {
        if (MFMA_ON) {
            // ---- MFMA-only wave: A tiles (taps, hi and lo) of all KT tap tiles in registers: 2 x KT x 4 VGPRs
            f16x8 ah[KT], al[KT];
#pragma unroll
            for (int kt = 0; kt < KT; ++kt)
#pragma unroll
                for (int j = 0; j < 8; ++j) { ah[kt][j] = (_Float16)(0.01f * (float)((lane + 3 * kt + j) & 31)); al[kt][j] = (_Float16)(0.001f * (float)((lane + kt + 5 * j) & 15)); }
            const int col = lane & 31, kg = lane >> 5;
            f32x16 acc;
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[e] = 0.0f;
            for (int it = 0; it < iters; ++it) {
                const int n0 = 32 * ((wave + it) % 50);                       // this wave's sample tile of the block
#pragma unroll
                for (int kt = 0; kt < KT; ++kt) {
                    const int e = n0 + col + 16 * kt + 8 * kg;               // first of the lane's eight consecutive samples
                    f16x8 bh, bl;
                    if (MODE == 0) {
                        // one 16-byte read at a 2-byte-aligned address (the compiler may not assume more: packed struct)
                        struct __attribute__((packed, aligned(2))) U { f16x8 v; };
                        bh = reinterpret_cast<const U*>(&xh[e])->v;
                        bl = reinterpret_cast<const U*>(&xl[e])->v;
                    } else {
                        const int sh = e & 7;                                 // copy `sh` is the window shifted left by sh samples
                        bh = *reinterpret_cast<const f16x8*>(&xh[sh * XW + (e - sh)]);
                        bl = *reinterpret_cast<const f16x8*>(&xl[sh * XW + (e - sh)]);
                    }
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[kt], bh, acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[kt], bh, acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[kt], bl, acc, 0, 0, 0);
                }
            }
#pragma unroll
            for (int e = 0; e < 16; ++e) sum += acc[e];
        }
    } else if (VALU_ON) {
        // ---- VALU-only wave: 32 independent accumulators, FMA chains (the issue pattern of the transform's butterflies)
        float v[32];
#pragma unroll
        for (int i = 0; i < 32; ++i) v[i] = (float)(lane + i) * 1e-3f;
        const float c0 = 1.0001f, c1 = 0.5f;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int rep = 0; rep < 9; ++rep)                                // 9 x 32 = 288 VALU instructions per iteration
#pragma unroll
                for (int i = 0; i < 32; ++i) v[i] = fmaf(v[i], c0, v[(i + 1) & 31] * 0.0f + c1);
        }
#pragma unroll
        for (int i = 0; i < 32; ++i) sum += v[i];
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    out[blockIdx.x * blockDim.x + tid] = sum;
    if (lane == 0) cyc[blockIdx.x * 12 + wave] = t1 - t0;
}

struct Res { double mfma_cyc, valu_cyc; };
template <int MODE, int NM, bool VALU_ON, bool MFMA_ON>
Res run(int iters) {
    float* out;
    unsigned long long* cyc;
    const int blocks = 256, threads = 768;
    (void)hipMalloc(&out, sizeof(float) * blocks * threads);
    (void)hipMalloc(&cyc, sizeof(unsigned long long) * blocks * 12);
    hipLaunchKernelGGL((k<MODE, NM, VALU_ON, MFMA_ON>), dim3(blocks), dim3(threads), 0, 0, out, cyc, 4);
    hipLaunchKernelGGL((k<MODE, NM, VALU_ON, MFMA_ON>), dim3(blocks), dim3(threads), 0, 0, out, cyc, iters);
    (void)hipDeviceSynchronize();
    std::vector<unsigned long long> h(blocks * 12);
    (void)hipMemcpy(h.data(), cyc, h.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost);
    double sm = 0, sv = 0;
    for (int b = 0; b < blocks; ++b)
        for (int w = 0; w < 12; ++w) (w < NM ? sm : sv) += (double)h[b * 12 + w];
    (void)hipFree(out);
    (void)hipFree(cyc);
    return {sm / (blocks * NM) / ((double)iters * KT * 3), sv / (blocks * (12 - NM)) / ((double)iters * 288)};
}

template <int MODE, int NM>
void report(const char* name) {
    const int iters = 300;
    const Res both = run<MODE, NM, true, true>(iters), m_only = run<MODE, NM, false, true>(iters), v_only = run<MODE, NM, true, false>(iters);
    const int vw = 12 - NM;                                                     // VALU waves per CU
    // cycles per MFMA of one wave -> MFMA pipe share of its SIMD: (waves per SIMD) x 32 / cycles
    const double wps = NM / 4.0;
    const double share_alone = wps * 32.0 / m_only.mfma_cyc, share_both = wps * 32.0 / both.mfma_cyc;
    // VALU instructions per cycle per SIMD: (VALU waves per SIMD) / cycles per instruction per wave
    const double ipc_alone = vw / 4.0 / v_only.valu_cyc, ipc_both = vw / 4.0 / both.valu_cyc;
    // one 1600-sample block: 50 sample tiles x KT tap tiles x 3 MFMAs over the CU's NM MFMA waves
    const double blk_cycles_alone = 50.0 * KT * 3 * m_only.mfma_cyc / NM, blk_cycles_both = 50.0 * KT * 3 * both.mfma_cyc / NM;
    printf("%s, %d MFMA wave(s) + %d VALU waves per CU:\n", name, NM, vw);
    printf("   MFMA waves alone: %6.1f cycles / MFMA / wave (pipe share %.2f);  beside the VALU waves: %6.1f (pipe share %.2f)\n", m_only.mfma_cyc,
           share_alone, both.mfma_cyc, share_both);
    printf("   VALU waves alone: %.3f instr / cycle / SIMD;  beside the MFMA waves: %.3f  (%.0f %% kept)\n", ipc_alone, ipc_both,
           100.0 * ipc_both / ipc_alone);
    printf("   one block (26 rows in one 32-row tile, %d taps, 3 products): %.1f us alone, %.1f us beside the VALU waves (2.4 GHz)\n", 16 * KT,
           blk_cycles_alone / 2.4e3, blk_cycles_both / 2.4e3);
    printf("{\"mode\": \"%s\", \"mfma_waves\": %d, \"pipe_share_alone\": %.3f, \"pipe_share_beside_valu\": %.3f, \"valu_kept\": %.3f, \"block_us_beside_valu\": %.2f}\n",
           name, NM, share_alone, share_both, ipc_both / ipc_alone, blk_cycles_both / 2.4e3);
}

int main() {
    printf("fp16-split Toeplitz MFMA waves (3 x v_mfma_f32_32x32x16_f16 + 2 x ds_read_b128 per (32-sample, 16-tap) tile; A tiles in registers)\n"
           "beside VALU-only waves on the same CU; 12-wave workgroups, one per CU.  Kill criterion (VERDICT r5 next #3): <= 4 us per block per CU.\n");
    report<0, 4>("unaligned Toeplitz reads");
    report<1, 4>("eight shifted copies (aligned reads)");
    report<0, 8>("unaligned Toeplitz reads");
    report<1, 8>("eight shifted copies (aligned reads)");
    return 0;
}
