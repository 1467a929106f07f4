// inst_fft_blkg_bwd_dx.hip -- instantiations of the run-time-geometry dL/dx kernel (leaf_fft_wgg_bwd.hpp).
// One of the translation units of libleaf_hip.so; see leaf_inst.hpp.
#define LEAF_INST_TU 1
#include "leaf_fft_wgg_bwd.hpp"
#include "leaf_inst.hpp"

const void* leaf_inst_fft_blkg_bwd_dx(int ni) {
    using K = void (*)(const FftParams);
    K fn = nullptr;
    switch (ni) {
        case 5: fn = leaf_fft_blkg_bwd_dx_kernel<5>; break;
        case 7: fn = leaf_fft_blkg_bwd_dx_kernel<7>; break;
        case 9: fn = leaf_fft_blkg_bwd_dx_kernel<9>; break;
        case 10: fn = leaf_fft_blkg_bwd_dx_kernel<10>; break;
        case 13: fn = leaf_fft_blkg_bwd_dx_kernel<13>; break;
        case 16: fn = leaf_fft_blkg_bwd_dx_kernel<16>; break;
        case 19: fn = leaf_fft_blkg_bwd_dx_kernel<19>; break;
    }
    return reinterpret_cast<const void*>(fn);
}

unsigned leaf_layout_fft_blkg_bwd_dx() { return leaf_layout_hash_fft(); }                // parameter-struct layout this unit was compiled with (leaf_inst.hpp)
