// log1p / expm1 for non-negative arguments on the hardware transcendental units (v_log_f32 / v_exp_f32, 1 ulp),
// used by the PCEN epilogue:  (q + d)^(1/r) - d^(1/r) = d^(1/r) * expm1(log1p(q/d) / r)  for d > 0.
// Both use Kahan's correction (evaluate the function at the ROUNDED intermediate u and rescale by the ratio of the
// exact argument to the one u stands for), so the relative accuracy holds for arbitrarily small arguments; measured
// against double precision by tools/check_fast_math.hip.
//
// `#pragma clang fp contract(off)` in every function: whether the compiler fuses `1.0f + q * inv_d` into one FMA depends on
// the code AROUND the inlined call, and the same frame finalized by two kernels (the row kernel, the workgroup kernels'
// streaming finalize) must come out bit-identical.
#pragma once
#include <hip/hip_runtime.h>

// x^y for x > 0 through the hardware log2/exp2: relative error about (1 + |y log2 x|) ulp.
__device__ __forceinline__ float leaf_pow_pos(float x, float y) {
#pragma clang fp contract(off)
    return __builtin_amdgcn_exp2f(y * __builtin_amdgcn_logf(x));
}

// (written as selects, not early returns: the compiler turns the returns into divergent branches, which cost more than the
// handful of instructions they skip and keep several calls in a row from being interleaved)
__device__ __forceinline__ float leaf_log1p_pos(float z) {        // z >= 0 (NaN propagates)
#pragma clang fp contract(off)
    const float u = 1.0f + z;
    const float d = u - 1.0f;                                      // the z that u represents exactly
    const float l = __builtin_amdgcn_logf(u) * 0.6931471805599453f;
    const float r = l * (z * __builtin_amdgcn_rcpf(d));            // ratio = 1 + O(eps): 1-ulp reciprocal is ample
    // d == 0: below half an ulp of 1, log1p(z) = z;  d == z: exact (includes huge z and +inf);  (d == 0 makes r inf or NaN: unused)
    return d == 0.0f ? z : (d == z ? l : r);
}

__device__ __forceinline__ float leaf_expm1_pos(float y) {        // y >= 0 (NaN propagates)
#pragma clang fp contract(off)
    const float u = __builtin_amdgcn_exp2f(y * 1.4426950408889634f);
    const float um1 = u - 1.0f;
    const float l = __builtin_amdgcn_logf(u) * 0.6931471805599453f;   // the y that u represents
    const float r = um1 * (y * __builtin_amdgcn_rcpf(l));
    // um1 == 0: e^y rounds to 1, expm1(y) = y;  um1 == u: huge (or +inf)
    return um1 == 0.0f ? y : (um1 == u ? u : r);
}
