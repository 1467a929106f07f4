// inst_fft_wg_bwd.hip -- instantiations of the static-geometry workgroup backward kernels (leaf_fft_wg_bwd.hpp).
// One of the translation units of libleaf_hip.so; see leaf_inst.hpp.
#define LEAF_INST_TU 1
#include "leaf_fft_wg_bwd.hpp"
#include "leaf_inst.hpp"

const void* leaf_inst_fft_wg_bwd(int sk) {
    using K = void (*)(const FftParams);
    K fn = nullptr;
    if (sk == 401) fn = leaf_fft_wg_bwd_kernel<401, 160, 12>;
    else if (sk == 801) fn = leaf_fft_wg_bwd_kernel<801, 320, 12>;
    else if (sk == 201) fn = leaf_fft_wg_bwd_kernel<201, 80, 12>;
    return reinterpret_cast<const void*>(fn);
}

const void* leaf_inst_fft_blk_bwd_dx(int sk) {
    using K = void (*)(const FftParams);
    K fn = nullptr;
    if (sk == 401) fn = leaf_fft_blk_bwd_dx_kernel<401, 160>;
    else if (sk == 801) fn = leaf_fft_blk_bwd_dx_kernel<801, 320>;
    else if (sk == 201) fn = leaf_fft_blk_bwd_dx_kernel<201, 80>;
    return reinterpret_cast<const void*>(fn);
}

unsigned leaf_layout_fft_wg_bwd() { return leaf_layout_hash_fft(); }                // parameter-struct layout this unit was compiled with (leaf_inst.hpp)
