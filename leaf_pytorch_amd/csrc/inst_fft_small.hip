// inst_fft_small.hip -- instantiations of the single-launch small-batch forward (leaf_fft_small.hpp).
// One of the translation units of libleaf_hip.so; see leaf_inst.hpp.
#define LEAF_INST_TU 1
#include "leaf_fft_small.hpp"
#include "leaf_inst.hpp"

// split: two workgroups per (clip, filter), seven waves each (grid (F, B, 2))
const void* leaf_inst_fft_small(int sk, bool split) {
    void (*fn)(const SmallParams) = nullptr;
    if (sk == 401) fn = split ? leaf_fft_small_kernel<401, 160, true> : leaf_fft_small_kernel<401, 160>;
    else if (sk == 201) fn = split ? leaf_fft_small_kernel<201, 80, true> : leaf_fft_small_kernel<201, 80>;
    return reinterpret_cast<const void*>(fn);
}

unsigned leaf_layout_fft_small() { return leaf_layout_hash_small(); }                // parameter-struct layout this unit was compiled with (leaf_inst.hpp)
