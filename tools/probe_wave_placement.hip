// Probe (GPU box): which SIMD of its CU each wave of a workgroup lands on, for the workgroup sizes the kernels use.
// HW_REG_HW_ID (gfx9 family): wave_id [3:0], simd_id [5:4], pipe_id [7:6], cu_id [11:8], sh_id [12], se_id [15:13].
//   hipcc --offload-arch=gfx950 -O2 tools/probe_wave_placement.hip -o /tmp/probe_wp && /tmp/probe_wp
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void k(unsigned* out, int spin) {
    unsigned id;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(id));
    float v = (float)threadIdx.x;
    for (int i = 0; i < spin; ++i) v = v * 1.0001f + 0.5f;            // keep the workgroup resident for a while
    if ((threadIdx.x & 63) == 0) out[blockIdx.x * 16 + (threadIdx.x >> 6)] = id | (v == 12345.f ? 1u << 31 : 0u);
}
int main() {
    const int blocks = 256;
    unsigned* d; hipMalloc(&d, blocks * 16 * 4);
    for (int threads : {512, 704, 768, 896, 1024}) {
        for (int lds : {0, 150 * 1024}) {
            hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            hipMemset(d, 0xff, blocks * 16 * 4);
            hipLaunchKernelGGL(k, dim3(blocks), dim3(threads), lds, 0, d, 20000);
            hipDeviceSynchronize();
            std::vector<unsigned> h(blocks * 16);
            hipMemcpy(h.data(), d, blocks * 16 * 4, hipMemcpyDeviceToHost);
            int hist[5][5] = {};                                        // hist[simd][count of waves on it] over workgroups
            int patterns[8] = {};
            for (int b = 0; b < blocks; ++b) {
                int per[4] = {};
                for (int w = 0; w < threads / 64; ++w) per[(h[b * 16 + w] >> 4) & 3]++;
                int mx = 0, mn = 99;
                for (int s = 0; s < 4; ++s) { mx = per[s] > mx ? per[s] : mx; mn = per[s] < mn ? per[s] : mn; }
                patterns[mx - mn]++;
                if (b < 2) printf("  threads %d lds %d block %d: waves per SIMD %d %d %d %d\n", threads, lds, b, per[0], per[1], per[2], per[3]);
            }
            printf("threads %4d lds %6d: workgroups by (max - min) waves per SIMD: 0:%d 1:%d 2:%d 3:%d 4:%d\n", threads, lds, patterns[0],
                   patterns[1], patterns[2], patterns[3], patterns[4]);
        }
    }
    return 0;
}
