// inst_fft_wgg4k_bwd.hip -- instantiations of the 4096-sample run-time-geometry backward kernel (leaf_fft_wgg4k_bwd.hpp).
// One of the translation units of libleaf_hip.so; see leaf_inst.hpp.
#define LEAF_INST_TU 1
#include "leaf_fft_wgg4k_bwd.hpp"
#include "leaf_inst.hpp"

const void* leaf_inst_fft_wgg4k_bwd(int ni2) {
    using K = void (*)(const FftParams);
    K fn = nullptr;
    switch (ni2) {
        case 10: fn = leaf_fft_wgg4k_bwd_kernel<12, 10>; break;
        case 13: fn = leaf_fft_wgg4k_bwd_kernel<12, 13>; break;
        case 17: fn = leaf_fft_wgg4k_bwd_kernel<12, 17>; break;
    }
    return reinterpret_cast<const void*>(fn);
}

// the static 32 kHz geometry (K = 801, hop = 320)
const void* leaf_inst_fft_wg4k_bwd() {
    using K = void (*)(const FftParams);
    K fn = leaf_fft_wgg4k_bwd_kernel<LEAF_4K_BWD_NW, 7, true>;
    return reinterpret_cast<const void*>(fn);
}

// ... with dL/dx: nine waves, the block's gradient spectra in LDS
const void* leaf_inst_fft_wg4k_bwd_dx() {
    using K = void (*)(const FftParams);
    K fn = leaf_fft_wgg4k_bwd_kernel<kWg4BwdDxWaves, 7, true, true>;
    return reinterpret_cast<const void*>(fn);
}

unsigned leaf_layout_fft_wgg4k_bwd() { return leaf_layout_hash_fft(); }                // parameter-struct layout this unit was compiled with (leaf_inst.hpp)
