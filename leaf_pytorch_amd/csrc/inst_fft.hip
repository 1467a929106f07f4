// inst_fft.hip -- instantiations of the per-wave overlap-save kernel, forward and backward (leaf_fft.hpp).
// One of the translation units of libleaf_hip.so; see leaf_inst.hpp.
#define LEAF_INST_TU 1
#include "leaf_fft.hpp"
#include "leaf_inst.hpp"

// sk = 401 | 801 | 201: static pooling geometry (real spectrum, double-buffered pooling row); sk = 0: run-time geometry with
// g2 = pooling-row buffers - 1 and rs = 1 (odd window) | 2 (even window: Hermitian K - 1 taps + the unpaired tap)
const void* leaf_inst_fft(int sk, int g2, int rs, int bwd) {
    using K = void (*)(const FftParams);
    K fn = nullptr;
    if (sk == 401) fn = bwd ? leaf_fft_kernel<401, 160, 1, 1, 1> : leaf_fft_kernel<401, 160, 1, 1, 0>;
    else if (sk == 801) fn = bwd ? leaf_fft_kernel<801, 320, 1, 1, 1> : leaf_fft_kernel<801, 320, 1, 1, 0>;
    else if (sk == 201) fn = bwd ? leaf_fft_kernel<201, 80, 1, 1, 1> : leaf_fft_kernel<201, 80, 1, 1, 0>;
    else if (sk == 0 && rs == 1)
        fn = bwd ? (g2 ? leaf_fft_kernel<0, 0, 1, 1, 1> : leaf_fft_kernel<0, 0, 0, 1, 1>)
                 : (g2 ? leaf_fft_kernel<0, 0, 1, 1, 0> : leaf_fft_kernel<0, 0, 0, 1, 0>);
    else if (sk == 0 && rs == 2)
        fn = bwd ? (g2 ? leaf_fft_kernel<0, 0, 1, 2, 1> : leaf_fft_kernel<0, 0, 0, 2, 1>)
                 : (g2 ? leaf_fft_kernel<0, 0, 1, 2, 0> : leaf_fft_kernel<0, 0, 0, 2, 0>);
    return reinterpret_cast<const void*>(fn);
}

unsigned leaf_layout_fft() { return leaf_layout_hash_fft(); }                // parameter-struct layout this unit was compiled with (leaf_inst.hpp)
