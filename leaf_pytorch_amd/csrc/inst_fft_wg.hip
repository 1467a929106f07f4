// inst_fft_wg.hip -- instantiations of the static-geometry workgroup forward kernels (leaf_fft_wg.hpp, leaf_fft_wg4k.hpp).
// One of the translation units of libleaf_hip.so; see leaf_inst.hpp.
#define LEAF_INST_TU 1
#include "leaf_fft_wg4k.hpp"
#include "leaf_inst.hpp"

// stream: the variant that finalizes in the kernel (whole clips per workgroup, frame sums in an LDS ring)
const void* leaf_inst_fft_wg(int sk, int nw, bool stream) {
    using K = void (*)(const FftParams);
    K fn = nullptr;
    if (stream) {
        if (sk == 401 && nw == 12) fn = leaf_fft_wg_kernel<401, 160, 12, true>;
        else if (sk == 801 && nw == 10) fn = leaf_fft_wg_kernel<801, 320, 10, true>;
        else if (sk == 201 && nw == 12) fn = leaf_fft_wg_kernel<201, 80, 12, true>;
        return reinterpret_cast<const void*>(fn);
    }
    if (sk == 401 && nw == 12) fn = leaf_fft_wg_kernel<401, 160, 12>;
    else if (sk == 801 && nw == 10) fn = leaf_fft_wg_kernel<801, 320, 10>;
    else if (sk == 201 && nw == 12) fn = leaf_fft_wg_kernel<201, 80, 12>;
#ifdef LEAF_WG_NW                   // A/B builds (tools/compare_builds.py name:-DLEAF_WG_NW=8): another workgroup size for 401/160
    else if (sk == 401 && nw == LEAF_WG_NW) fn = leaf_fft_wg_kernel<401, 160, LEAF_WG_NW>;
#endif
#if LEAF_TOOLS                      // LEAF_WG_WAVES=16: the column-half transposition form (A/B measurements)
    else if (sk == 401 && nw == 16) fn = leaf_fft_wg_kernel<401, 160, 16>;
    else if (sk == 801 && nw == 14) fn = leaf_fft_wg_kernel<801, 320, 14>;
    else if (sk == 201 && nw == 16) fn = leaf_fft_wg_kernel<201, 80, 16>;
#endif
    return reinterpret_cast<const void*>(fn);
}

const void* leaf_inst_fft_wg4k() {
    void (*fn)(const FftParams) = leaf_fft_wg4k_kernel<801, 320, LEAF_4K_FWD_NW>;
    return reinterpret_cast<const void*>(fn);
}

unsigned leaf_layout_fft_wg() { return leaf_layout_hash_fft(); }                // parameter-struct layout this unit was compiled with (leaf_inst.hpp)
