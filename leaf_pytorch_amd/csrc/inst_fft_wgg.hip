// inst_fft_wgg.hip -- instantiations of the run-time-geometry workgroup forward kernels (leaf_fft_wgg.hpp, leaf_fft_wgg4k.hpp).
// One of the translation units of libleaf_hip.so; see leaf_inst.hpp.
#define LEAF_INST_TU 1
#include "leaf_fft_wgg4k.hpp"
#include "leaf_inst.hpp"

// ni = taps per lane (bucketed 5/7/9/10/13/16/19); the full transposition scratch (half_scratch = false) exists up to 10
const void* leaf_inst_fft_wgg(int ni, bool half_scratch) {
    using K = void (*)(const FftParams);
    K fn = nullptr;
    if (!half_scratch) {
        switch (ni) {
            case 5: fn = leaf_fft_wgg_kernel<12, 5, false>; break;
            case 7: fn = leaf_fft_wgg_kernel<12, 7, false>; break;
            case 9: fn = leaf_fft_wgg_kernel<12, 9, false>; break;
            case 10: fn = leaf_fft_wgg_kernel<12, 10, false>; break;
        }
    } else {
        switch (ni) {
            case 5: fn = leaf_fft_wgg_kernel<12, 5>; break;
            case 7: fn = leaf_fft_wgg_kernel<12, 7>; break;
            case 9: fn = leaf_fft_wgg_kernel<12, 9>; break;
            case 10: fn = leaf_fft_wgg_kernel<12, 10>; break;
            case 13: fn = leaf_fft_wgg_kernel<12, 13>; break;
            case 16: fn = leaf_fft_wgg_kernel<12, 16>; break;
            case 19: fn = leaf_fft_wgg_kernel<12, 19>; break;
        }
    }
    return reinterpret_cast<const void*>(fn);
}

// ni2 = taps of one parity per lane on the 4096-sample plan (bucketed 10/13/17)
const void* leaf_inst_fft_wgg4k(int ni2) {
    using K = void (*)(const FftParams);
    K fn = nullptr;
    switch (ni2) {
        case 10: fn = leaf_fft_wgg4k_kernel<12, 10>; break;
        case 13: fn = leaf_fft_wgg4k_kernel<12, 13>; break;
        case 17: fn = leaf_fft_wgg4k_kernel<12, 17>; break;
    }
    return reinterpret_cast<const void*>(fn);
}

unsigned leaf_layout_fft_wgg() { return leaf_layout_hash_fft(); }                // parameter-struct layout this unit was compiled with (leaf_inst.hpp)
