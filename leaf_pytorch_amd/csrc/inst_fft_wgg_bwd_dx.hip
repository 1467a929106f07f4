// inst_fft_wgg_bwd_dx.hip -- instantiations of the run-time-geometry workgroup backward kernel that also yields dL/dx
// (leaf_fft_wgg_bwd.hpp, DX = true).  One of the translation units of libleaf_hip.so; see leaf_inst.hpp.
#define LEAF_INST_TU 1
#include "leaf_fft_wgg_bwd.hpp"
#include "leaf_inst.hpp"

const void* leaf_inst_fft_wgg_bwd_dx(int ni) {
    using K = void (*)(const FftParams);
    K fn = nullptr;
    switch (ni) {
        case 5: fn = leaf_fft_wgg_bwd_kernel<12, 5, true, true>; break;
        case 7: fn = leaf_fft_wgg_bwd_kernel<12, 7, true, true>; break;
        case 9: fn = leaf_fft_wgg_bwd_kernel<12, 9, true, true>; break;
        case 10: fn = leaf_fft_wgg_bwd_kernel<12, 10, true, true>; break;
        case 13: fn = leaf_fft_wgg_bwd_kernel<12, 13, true, true>; break;
        case 16: fn = leaf_fft_wgg_bwd_kernel<12, 16, true, true>; break;
        case 19: fn = leaf_fft_wgg_bwd_kernel<12, 19, true, true>; break;
    }
    return reinterpret_cast<const void*>(fn);
}

unsigned leaf_layout_fft_wgg_bwd_dx() { return leaf_layout_hash_fft(); }                // parameter-struct layout this unit was compiled with (leaf_inst.hpp)
