// leaf_inst.hpp -- handles of the big kernel-template instantiations, one getter per kernel family.
//
// libleaf_hip.so is built from several translation units compiled in parallel (leaf_pytorch_amd/_native.py): leaf_kernels.hip
// holds the C ABI, the host logic and the small kernels; each inst_*.hip instantiates one family of the big templates and
// hands out their host-side handles through the getters below.  A getter returns nullptr for a combination that is not
// instantiated (the callers treat that as "this kernel does not exist for the geometry").  The handles are opaque because
// every kernel takes its parameter struct by value and those structs live in the headers' unnamed namespaces; the caller
// casts to `void (*)(const FftParams)` etc. -- same definition, same ABI, different translation unit.
//
// LEAF_TOOLS (compile-time, default 0): measurement builds additionally instantiate the A/B variants the tools/ scripts
// select through environment variables (16-wave static kernels, wide NOFF = 6 register tiles of the MFMA kernel).
#pragma once
#ifndef LEAF_TOOLS
#define LEAF_TOOLS 0
#endif

const void* leaf_inst_fused(int rt, int noff, bool even_k, bool bwd);          // leaf_fused_kernel<RT, NOFF, EVENK, BWD>
const void* leaf_inst_dtaps(int rt, int tpw, bool even_k);                     // dtaps_mfma_kernel<RT, TPW, EVENK>
const void* leaf_inst_fft(int sk, int g2, int rs, int bwd);                    // leaf_fft_kernel<SK, SHOP, G2, RS, BWD>; sk = 0 | 201 | 401 | 801
const void* leaf_inst_fft_wg(int sk, int nw, bool stream);                     // leaf_fft_wg_kernel<SK, SHOP, NW, STREAM>
const void* leaf_inst_fft_small(int sk, bool split);                          // leaf_fft_small_kernel<SK, SHOP, SPLIT>: sk = 401 | 201
const void* leaf_inst_fft_wg4k();                                              // leaf_fft_wg4k_kernel<801, 320, 12>
const void* leaf_inst_fft_wgg(int ni, bool half_scratch);                      // leaf_fft_wgg_kernel<12, NI, HALF>
const void* leaf_inst_fft_wgg4k(int ni2);                                      // leaf_fft_wgg4k_kernel<12, NI2>
const void* leaf_inst_fft_wg_bwd(int sk);                                      // leaf_fft_wg_bwd_kernel<SK, SHOP, 12>
const void* leaf_inst_fft_wg_bwd_dx(int sk);                                   // leaf_fft_wg_bwd_kernel<SK, SHOP, 12, true>: + dL/dx (K = 401, 201)
const void* leaf_inst_fft_blk_bwd_dx(int sk);                                  // leaf_fft_blk_bwd_dx_kernel<SK, SHOP>
const void* leaf_inst_fft_wgg_bwd(int ni, bool half_scratch);                  // leaf_fft_wgg_bwd_kernel<12, NI, HALF>
const void* leaf_inst_fft_wgg_bwd_dx(int ni);                                  // leaf_fft_wgg_bwd_kernel<12, NI, true, true>: + dL/dx
const void* leaf_inst_fft_wgg4k_bwd(int ni2);                                  // leaf_fft_wgg4k_bwd_kernel<12, NI2>
const void* leaf_inst_fft_wg4k_bwd();                                          // leaf_fft_wgg4k_bwd_kernel<12, 7, true>: K = 801, hop = 320
const void* leaf_inst_fft_wg4k_bwd_dx();                                       // leaf_fft_wgg4k_bwd_kernel<9, 7, true, true>: the same with dL/dx

// Parameter-struct layout fingerprint of each unit (leaf_layout_hash_* of leaf_fused.hpp / leaf_fft.hpp / ...): compared by
// leaf_kernels.hip with the fingerprint of ITS copy of the structs before the first launch (inst_layouts_ok).
unsigned leaf_layout_fft();
unsigned leaf_layout_fft_small();
unsigned leaf_layout_fft_wg();
unsigned leaf_layout_fft_wg_bwd();
unsigned leaf_layout_fft_wg_bwd_dx();
unsigned leaf_layout_fft_wgg();
unsigned leaf_layout_fft_wgg4k_bwd();
unsigned leaf_layout_fft_wgg_bwd();
unsigned leaf_layout_fft_wgg_bwd_dx();
unsigned leaf_layout_fused();
