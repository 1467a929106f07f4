// Microbenchmark (GPU box): issue cost of scalar vs packed fp32 VALU instructions on gfx950, measured with HIP events.
// One wave per SIMD (256-thread workgroups, one per CU) and two waves per SIMD; 16 independent dependency chains.
//   hipcc --offload-arch=gfx950 -O3 tools/ubench_valu.hip -o /tmp/ubench_valu && /tmp/ubench_valu
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x2 __attribute__((ext_vector_type(2)));

template <int MODE>
__global__ __launch_bounds__(512) void k(float* out, int iters) {
    const int lane = threadIdx.x;
    f32x2 v[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) v[i] = f32x2{(float)(lane + i), (float)(lane - i)};
    const f32x2 m = {1.0001f, 0.9999f}, c = {0.5f, -0.5f};
    const unsigned long long cond = 0x5555555555555555ull + (unsigned long long)iters;
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
                }
            }
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) s += v[i].x + v[i].y;
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int MODE>
void run(const char* name, int threads, int instr_per_slot) {
    const int iters = 20000, blocks = 256;
    float* out; hipMalloc(&out, blocks * 512 * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float best = 1e9f;
    for (int rep = 0; rep < 4; ++rep) {
        hipEventRecord(e0, 0);
        hipLaunchKernelGGL((k<MODE>), dim3(blocks), dim3(threads), 0, 0, out, iters);
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

int main() {
    for (int t : {256, 512}) {
        run<0>("2x v_fma_f32", t, 2);
        run<4>("v_fma_f32", t, 1);
        run<5>("v_add_f32", t, 1);
        run<1>("v_pk_fma_f32", t, 1);
        run<2>("v_pk_mul_f32", t, 1);
        run<3>("v_pk_add_f32", t, 1);
        run<6>("v_permlane32_swap_b32", t, 1);
        run<7>("v_cndmask_b32_e64", t, 1);
    }
    return 0;
}
