// inst_fft_wgg_bwd.hip -- instantiations of the run-time-geometry workgroup backward kernels (leaf_fft_wgg_bwd.hpp).
// One of the translation units of libleaf_hip.so; see leaf_inst.hpp.
#define LEAF_INST_TU 1
#include "leaf_fft_wgg_bwd.hpp"
#include "leaf_inst.hpp"

const void* leaf_inst_fft_wgg_bwd(int ni, bool half_scratch) {
    using K = void (*)(const FftParams);
    K fn = nullptr;
    if (!half_scratch) {
        switch (ni) {
            case 5: fn = leaf_fft_wgg_bwd_kernel<12, 5, false>; break;
            case 7: fn = leaf_fft_wgg_bwd_kernel<12, 7, false>; break;
            case 9: fn = leaf_fft_wgg_bwd_kernel<12, 9, false>; break;
            case 10: fn = leaf_fft_wgg_bwd_kernel<12, 10, false>; break;
        }
    } else {
        switch (ni) {
            case 5: fn = leaf_fft_wgg_bwd_kernel<12, 5>; break;
            case 7: fn = leaf_fft_wgg_bwd_kernel<12, 7>; break;
            case 9: fn = leaf_fft_wgg_bwd_kernel<12, 9>; break;
            case 10: fn = leaf_fft_wgg_bwd_kernel<12, 10>; break;
            case 13: fn = leaf_fft_wgg_bwd_kernel<12, 13>; break;
            case 16: fn = leaf_fft_wgg_bwd_kernel<12, 16>; break;
            case 19: fn = leaf_fft_wgg_bwd_kernel<12, 19>; break;
        }
    }
    return reinterpret_cast<const void*>(fn);
}

unsigned leaf_layout_fft_wgg_bwd() { return leaf_layout_hash_fft(); }                // parameter-struct layout this unit was compiled with (leaf_inst.hpp)
