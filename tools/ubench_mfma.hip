// Microbenchmark (GPU box): does v_mfma_f32_16x16x4_f32 overlap with VALU / LDS instructions of the same wave and of
// a second wave on the same SIMD?  Prints cycles per MFMA (s_memtime) for: MFMA only; MFMA + K independent v_fma per
// MFMA; MFMA + K ds_read_b32 per MFMA -- with 1 and 2 waves per SIMD.
//   hipcc --offload-arch=gfx950 -O3 tools/ubench_mfma.hip -o /tmp/ubench_mfma && /tmp/ubench_mfma
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int KV, int KL>
__global__ __launch_bounds__(512) void k(float* out, unsigned long long* cyc, int iters) {
    __shared__ float lds[4096];
    const int lane = threadIdx.x & 63;
    for (int i = threadIdx.x; i < 4096; i += blockDim.x) lds[i] = (float)i * 1e-3f;
    __syncthreads();
    f32x4 acc[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    float a = (float)lane * 0.01f, b = 1.0f + (float)lane * 1e-3f;
    float v[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = (float)(lane + i);
    float l[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) l[i] = 0.f;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
#pragma unroll
            for (int j = 0; j < KV; ++j) v[(i + j) & 7] = fmaf(v[(i + j) & 7], 1.0001f, 0.5f);
#pragma unroll
            for (int j = 0; j < KL; ++j) l[(i + j) & 7] += lds[(lane + 64 * ((i + j) & 31) + it) & 4095];
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
#pragma unroll
    for (int i = 0; i < 8; ++i) s += v[i] + l[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (lane == 0) cyc[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0;
}

template <int KV, int KL>
void run(const char* name, int waves_per_wg) {
    const int iters = 2000, blocks = 256;
    float* out; unsigned long long* cyc;
    hipMalloc(&out, blocks * 512 * 4); hipMalloc(&cyc, blocks * 8 * 8);
    for (int rep = 0; rep < 2; ++rep) {
        hipLaunchKernelGGL((k<KV, KL>), dim3(blocks), dim3(64 * waves_per_wg), 0, 0, out, cyc, iters);
        hipDeviceSynchronize();
    }
    std::vector<unsigned long long> h(blocks * waves_per_wg);
    hipMemcpy(h.data(), cyc, h.size() * 8, hipMemcpyDeviceToHost);
    double sum = 0; for (auto c : h) sum += (double)c;
    const double per_wave = sum / h.size() / (iters * 16.0);
    printf("%-28s waves/SIMD=%d  cycles per MFMA per wave = %7.2f   -> SIMD time per MFMA = %6.2f\n", name,
           waves_per_wg / 4, per_wave, per_wave / (waves_per_wg / 4));
    hipFree(out); hipFree(cyc);
}

int main() {
    for (int w : {4, 8}) {
        run<0, 0>("mfma only", w);
        run<1, 0>("mfma + 1 v_fma", w);
        run<2, 0>("mfma + 2 v_fma", w);
        run<4, 0>("mfma + 4 v_fma", w);
        run<0, 1>("mfma + 1 ds_read", w);
        run<0, 2>("mfma + 2 ds_read", w);
        run<1, 1>("mfma + 1 v_fma + 1 ds_read", w);
    }
    return 0;
}
