// Accuracy check (GPU box) of the hardware-transcendental log1p / expm1 used by the PCEN epilogue, against double
// precision, over log-spaced arguments.  Prints the maximum relative error of each.
//   hipcc --offload-arch=gfx950 -O3 -I leaf_pytorch_amd/csrc tools/check_fast_math.hip -o /tmp/check_fast_math && /tmp/check_fast_math
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <vector>
#include "leaf_fastmath.hpp"

__global__ void k(const float* x, float* l1p, float* em1, float* pw, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    l1p[i] = leaf_log1p_pos(x[i]);
    em1[i] = leaf_expm1_pos(x[i]);
    pw[i] = 1.41421356f * leaf_expm1_pos(0.5f * leaf_log1p_pos(x[i] * 0.5f));     // (x+2)^0.5 - 2^0.5
}

int main() {
    const int n = 1 << 20;
    std::vector<float> x(n);
    for (int i = 0; i < n; ++i) x[i] = std::pow(10.0, -12.0 + 18.0 * i / (n - 1.0));   // 1e-12 .. 1e6
    float *dx, *d1, *d2, *d3;
    hipMalloc(&dx, n * 4); hipMalloc(&d1, n * 4); hipMalloc(&d2, n * 4); hipMalloc(&d3, n * 4);
    hipMemcpy(dx, x.data(), n * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(n / 256), dim3(256), 0, 0, dx, d1, d2, d3, n);
    std::vector<float> a(n), b(n), c(n);
    hipMemcpy(a.data(), d1, n * 4, hipMemcpyDeviceToHost);
    hipMemcpy(b.data(), d2, n * 4, hipMemcpyDeviceToHost);
    hipMemcpy(c.data(), d3, n * 4, hipMemcpyDeviceToHost);
    double e1 = 0, e2 = 0, e3 = 0, x1 = 0, x2 = 0, x3 = 0;
    for (int i = 0; i < n; ++i) {
        const double xd = x[i];
        const double r1 = std::log1p(xd), r3 = std::sqrt(xd + 2.0) - std::sqrt(2.0);
        const double q1 = std::fabs(a[i] - r1) / r1, q3 = std::fabs(c[i] - std::sqrt(2.0) * std::expm1(0.5 * std::log1p(xd * 0.5))) /
                                                       (std::sqrt(2.0) * std::expm1(0.5 * std::log1p(xd * 0.5)));
        (void)r3;
        if (q1 > e1) { e1 = q1; x1 = xd; }
        if (q3 > e3) { e3 = q3; x3 = xd; }
        if (xd < 80.0) {
            const double r2 = std::expm1(xd), q2 = std::fabs(b[i] - r2) / r2;
            if (q2 > e2) { e2 = q2; x2 = xd; }
        }
    }
    printf("log1p  max rel err %.3e at x=%.3e\nexpm1  max rel err %.3e at x=%.3e\npcen   max rel err %.3e at q=%.3e\n", e1, x1, e2,
           x2, e3, x3);
    return 0;
}
