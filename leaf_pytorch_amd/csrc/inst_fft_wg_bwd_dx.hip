// inst_fft_wg_bwd_dx.hip -- instantiations of the static-geometry workgroup backward kernel that also yields dL/dx
// (leaf_fft_wg_bwd.hpp, DX = true).  One of the translation units of libleaf_hip.so; see leaf_inst.hpp.
#define LEAF_INST_TU 1
#include "leaf_fft_wg_bwd.hpp"
#include "leaf_inst.hpp"

const void* leaf_inst_fft_wg_bwd_dx(int sk) {
    using K = void (*)(const FftParams);
    K fn = nullptr;
    if (sk == 401) fn = leaf_fft_wg_bwd_kernel<401, 160, 12, true>;
    else if (sk == 201) fn = leaf_fft_wg_bwd_kernel<201, 80, 12, true>;
    return reinterpret_cast<const void*>(fn);
}

unsigned leaf_layout_fft_wg_bwd_dx() { return leaf_layout_hash_fft(); }                // parameter-struct layout this unit was compiled with (leaf_inst.hpp)
