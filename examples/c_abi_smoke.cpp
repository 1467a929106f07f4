// The drop-in boundary without Python or torch: a plain C++ host binds include/leaf_hip.h, runs the whole LEAF forward
// on a synthetic batch through LEAF_ALGO_AUTO and checks it against the same library's staged per-module kernels
// (one HIP kernel per reference module, materialising every intermediate like the reference graph).
//   g++ -O2 -D__HIP_PLATFORM_AMD__ -I /opt/rocm/include -I include examples/c_abi_smoke.cpp -L leaf_pytorch_amd -lleaf_hip \
//       -L /opt/rocm/lib -lamdhip64 -Wl,-rpath,$PWD/leaf_pytorch_amd -Wl,-rpath,/opt/rocm/lib -o /tmp/c_abi_smoke      (or hipcc: no device code here)
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "leaf_hip.h"

#define HIP_OK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); return 2; } } while (0)
#define LEAF_OK_(x) do { int rc_ = (x); if (rc_ != LEAF_OK) { printf("leaf error %d (%s) at %s:%d\n", rc_, leaf_status_string(rc_), __FILE__, __LINE__); return 3; } } while (0)

int main() {
    const int B = 8, T = 16000, F = 40, K = 401, hop = 160;
    const int TP = leaf_num_frames(T, K, hop);
    std::vector<float> x((size_t)B * T), kernel(2 * F), pool_w(F, 0.4f), pool_b(F, 1.0f), alpha(F, 0.96f), delta(F, 2.0f),
        root(F, 2.0f), ema_w(F, 0.04f);
    unsigned s = 12345u;
    for (auto& v : x) { s = s * 1664525u + 1013904223u; v = (float)(s >> 8) / 8388608.0f - 1.0f; }
    for (int f = 0; f < F; ++f) {                      // centre frequencies spread over (0, pi), widths 10..90 samples
        kernel[2 * f] = 0.05f + 2.9f * (float)f / (float)(F - 1);
        kernel[2 * f + 1] = 90.0f - 80.0f * (float)f / (float)(F - 1);
    }
    auto upload = [](const std::vector<float>& h, float** d) {
        if (hipMalloc(d, h.size() * 4) != hipSuccess) return false;
        return hipMemcpy(*d, h.data(), h.size() * 4, hipMemcpyHostToDevice) == hipSuccess;
    };
    float *dx, *dk, *dpw, *dpb, *da, *dd, *dr, *dw, *dout, *dref;
    if (!upload(x, &dx) || !upload(kernel, &dk) || !upload(pool_w, &dpw) || !upload(pool_b, &dpb) || !upload(alpha, &da) ||
        !upload(delta, &dd) || !upload(root, &dr) || !upload(ema_w, &dw)) return 2;
    HIP_OK(hipMalloc(&dout, (size_t)B * F * TP * 4));
    HIP_OK(hipMalloc(&dref, (size_t)B * F * TP * 4));
    hipStream_t st;
    HIP_OK(hipStreamCreate(&st));

    const int algo = leaf_auto_algo(B, T, F, K, hop);
    const size_t ws_bytes = leaf_workspace_bytes(B, T, F, K, hop, LEAF_ALGO_AUTO);
    void* ws;
    HIP_OK(hipMalloc(&ws, ws_bytes));
    LEAF_OK_(leaf_forward_f32(dx, B, T, dk, dpw, dpb, da, dd, dr, dw, F, K, hop, LEAF_FLAG_PCEN, LEAF_ALGO_AUTO, dout, ws,
                              ws_bytes, st));

    // reference: the staged kernels, module by module (convolution.py:71-99, frontend.py:15-19, pooling.py:31-42, :84,
    // postprocessing.py:62-69)
    float *dy, *de, *dp, *dtaps, *dg;
    HIP_OK(hipMalloc(&dy, (size_t)B * 2 * F * T * 4));
    HIP_OK(hipMalloc(&de, (size_t)B * F * T * 4));
    HIP_OK(hipMalloc(&dp, (size_t)B * F * TP * 4));
    HIP_OK(hipMalloc(&dtaps, (size_t)2 * F * K * 4));
    HIP_OK(hipMalloc(&dg, (size_t)F * K * 4));
    LEAF_OK_(leaf_gabor_conv_f32(dx, B, T, dk, F, K, dy, dtaps, (size_t)2 * F * K * 4, st));
    LEAF_OK_(leaf_squared_modulus_f32(dy, B, F, T, de, st));
    LEAF_OK_(leaf_gaussian_lowpass_f32(de, B, F, T, dpw, dpb, K, hop, dp, dg, (size_t)F * K * 4, st));
    // (frontend.py:84's 1e-5 floor is inert here: pooled >= bias = 1)
    LEAF_OK_(leaf_pcen_f32(dp, B, F, TP, da, dd, dr, dw, 1e-12f, dref, st));
    HIP_OK(hipStreamSynchronize(st));

    std::vector<float> out((size_t)B * F * TP), ref(out.size());
    HIP_OK(hipMemcpy(out.data(), dout, out.size() * 4, hipMemcpyDeviceToHost));
    HIP_OK(hipMemcpy(ref.data(), dref, ref.size() * 4, hipMemcpyDeviceToHost));
    double worst = 0.0;
    for (size_t i = 0; i < out.size(); ++i) {
        if (!std::isfinite(out[i])) { printf("non-finite output at %zu\n", i); return 4; }
        worst = std::fmax(worst, std::fabs((double)out[i] - ref[i]) / std::fmax(std::fabs((double)ref[i]), 1e-30));
    }
    printf("c_abi_smoke: abi %d, algo %d, (%d,1,%d) -> (%d,%d,%d), max rel diff fused vs staged %.3e\n", leaf_abi_version(), algo,
           B, T, B, F, TP, worst);
    return worst < 2e-5 ? 0 : 5;
}
