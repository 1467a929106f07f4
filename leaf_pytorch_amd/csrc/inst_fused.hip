// inst_fused.hip -- instantiations of the direct-form MFMA kernels (leaf_fused.hpp, leaf_backward.hpp).
// One of the translation units of libleaf_hip.so; see leaf_inst.hpp.
#define LEAF_INST_TU 1
#include "leaf_backward.hpp"
#include "leaf_inst.hpp"

namespace {
template <int RT, int NOFF>
const void* fused_pick(bool even_k, bool bwd) {
    void (*fn)(const FusedParams) = even_k ? (bwd ? leaf_fused_kernel<RT, NOFF, true, true> : leaf_fused_kernel<RT, NOFF, true, false>)
                                           : (bwd ? leaf_fused_kernel<RT, NOFF, false, true> : leaf_fused_kernel<RT, NOFF, false, false>);
    return reinterpret_cast<const void*>(fn);
}
template <int RT, int TPW>
const void* dtaps_pick(bool even_k) {
    void (*fn)(const DtapsParams) = even_k ? dtaps_mfma_kernel<RT, TPW, true> : dtaps_mfma_kernel<RT, TPW, false>;
    return reinterpret_cast<const void*>(fn);
}
}  // namespace

// NOFF = 6 (4..6 overlapping frames per hop-block) exists at RT = 1 only: the wider register tiles spill (720 / 188 bytes per
// lane at RT = 3 / 2) and are slower than RT = 1 as well (tools/tune_rt_cap.py); LEAF_TOOLS builds keep them for that sweep.
const void* leaf_inst_fused(int rt, int noff, bool even_k, bool bwd) {
    switch (rt * 10 + noff) {
        case 11: return fused_pick<1, 1>(even_k, bwd);
        case 13: return fused_pick<1, 3>(even_k, bwd);
        case 16: return fused_pick<1, 6>(even_k, bwd);
        case 21: return fused_pick<2, 1>(even_k, bwd);
        case 23: return fused_pick<2, 3>(even_k, bwd);
        case 31: return fused_pick<3, 1>(even_k, bwd);
        case 33: return fused_pick<3, 3>(even_k, bwd);
#if LEAF_TOOLS
        case 26: return fused_pick<2, 6>(even_k, bwd);
        case 36: return fused_pick<3, 6>(even_k, bwd);
#endif
    }
    return nullptr;
}

const void* leaf_inst_dtaps(int rt, int tpw, bool even_k) {
    switch (rt * 10 + tpw) {
        case 11: return dtaps_pick<1, 1>(even_k);
        case 12: return dtaps_pick<1, 2>(even_k);
        case 13: return dtaps_pick<1, 3>(even_k);
        case 21: return dtaps_pick<2, 1>(even_k);
        case 22: return dtaps_pick<2, 2>(even_k);
        case 23: return dtaps_pick<2, 3>(even_k);
        case 31: return dtaps_pick<3, 1>(even_k);
        case 32: return dtaps_pick<3, 2>(even_k);
        case 33: return dtaps_pick<3, 3>(even_k);
    }
    return nullptr;
}

unsigned leaf_layout_fused() { return leaf_layout_hash_bwd(); }                // parameter-struct layout this unit was compiled with (leaf_inst.hpp)
