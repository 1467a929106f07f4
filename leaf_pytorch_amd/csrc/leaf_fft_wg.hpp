// leaf_fft_wg.hpp -- overlap-save forward, second generation: one WORKGROUP per block, spectrum shared through LDS
// Part of the single translation unit leaf_kernels.hip (gfx950 only); see that file's header comment.
//
// Why (round-1 profile of leaf_fft_kernel, profiles/r01): the kernel is fp32-VALU-issue-bound but issues only ~0.49 of
// the SIMD's slots: a wave issues at most one VALU instruction per ~4.6 cycles, so two waves per SIMD cannot fill a
// 2-cycle pipe, and each wave-level FFT has four dependent LDS phases.  Two waves per SIMD was forced by registers: the
// block's spectrum A' (64 VGPRs) stayed resident across the wave's filters next to the 64 registers of the transform.
//
// Here A' lives in LDS instead, computed ONCE per block and read by every wave of the workgroup at the spectral
// multiply, which (i) frees 64 + 32 VGPRs -> three waves per SIMD (12-wave workgroups, <= 168 VGPRs), (ii) removes the
// repeated forward transforms (one per block instead of one per (block, filter group)), and (iii) needs only half the
// spectrum: the (rotated) block is real, so A'[N - k] = conj(A'[k]) and bins 0..1024 are stored (8.2 KB per block);
// the upper half is read back mirrored (a descending ds_read_b64, conflict-free like the ascending one).
//
// Scheduling: no barriers.  A workgroup walks its blocks (blockIdx.x, + gridDim.x, ...) through a task queue in LDS:
//     fwd(0), [fwd(1), inv(0,0) .. inv(0,F-1)], [fwd(2), inv(1,0) ..], ...
// pulled in order with one LDS atomic per task.  fwd(i) = load + forward transform of set i's block into ring slot i & 1;
// inv(i,f) = spectral multiply with filter f + inverse transform + |.|^2 + pooling (the arithmetic of
// leaf_fft_kernel's static-geometry path; results differ from it only by the rounding of the mirrored upper half-spectrum,
// ~1e-7 relative, so a clip is bit-identical across batches served by THIS kernel, not across the two kernels).  Dependencies are two monotonic counters per slot:
//     inv(i,f) waits for  fwd_cnt[slot] >= (i >> 1) + 1          (the spectrum is there)
//     fwd(i)   waits for  inv_cnt[slot] >= (i >> 1) * F          (every reader of the slot's previous occupant is done)
// Because fwd(i+1) is queued BEFORE set i's inverse tasks, one wave computes the next spectrum while the other eleven
// work on the current block, and nobody waits for it.  Deadlock-free: a task only ever waits for tasks queued before it.
#pragma once
#include <type_traits>
#include "leaf_fft.hpp"

namespace {

constexpr int kWgRingFloat2 = 1032;            // bins 0..1024 of a block's spectrum, padded
constexpr int kWgFwdBins = 1152;               // static FORWARD kernel (leaf_fft_wg_kernel) and the first-block spectra of the table launch: bins 0..1151 -- the
constexpr int kWgRingFwdFloat2 = kWgFwdBins + 8;   // transform yields all 2048 bins, and band tasks whose window crosses Nyquist (leaf_band.hpp, round 6) read bins
                                               // kb .. kb + M - 1 < 1152 as they lie (bins above 1024 are the conjugate mirror of a real block's spectrum).  128 bins
                                               // beyond Nyquist, not 256: with 2 x 2 KB more the streaming-finalize instance would fall from a 64-frame ring to 32
                                               // (lag 6 -> 2, +5 % on BASELINE configs[3] / [4]); the default bank's top filter needs 111
constexpr int kWgQueueInts = 16;               // q_next, fwd_cnt[2], inv_cnt[2], {clip, block-in-clip} per ring slot, [11..12] blocks
                                               // finalized per parity (STREAM kernels; their per-block counters live behind their ring)

// The queue protocol between the waves of a workgroup: data (ring slots, block coordinates, shared sums) is written with plain LDS
// stores, then a counter moves (relaxed atomic add by lane 0) and the readers spin on it (relaxed atomic loads).  The ordering
// is carried by FENCES restricted to the LDS address space: release before the counter moves, acquire after the spin.  On
// gfx950 the release fence is the s_waitcnt lgkmcnt(0) these sites issued by hand in round 2 and the acquire fence emits no
// instruction (a wave's LDS operations execute in order) -- the generated code is unchanged -- but the compiler now KNOWS it
// may not move LDS accesses across them; the unrestricted fences would also wait for the global loads in flight (table
// prefetches), which the protocol does not need.
__device__ __forceinline__ int wg_ld(const int* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
__device__ __forceinline__ void wg_release() { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local"); }
__device__ __forceinline__ void wg_wait_ge(const int* p, int need) {
    while (wg_ld(p) < need) __builtin_amdgcn_s_sleep(1);
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}

// ---- wave-level 2048-point FFT, LDS-lean variant of fft2048 (same arithmetic, same register conventions) ----------
// At three waves per SIMD the LDS pipe (one per CU) is as busy as the VALUs (profiles/r02), so the transposes and table
// reads are re-shaped around the LDS cycle table of MI355X_MICROARCH.md:
//   * transposition writes: ds_write_addtid_b32 (address = M0 + offset + 4 lane, no address VGPR): 2 LDS cycles per
//     wave-instruction instead of ds_write_b32's 4 -- the write IS "row brev5(i), column lane", exactly the add-tid form;
//   * transposition reads: a lane's 32 values are contiguous in its row, so with a row stride of 68 floats (16-byte
//     aligned rows; the four 16-lane groups of a ds_read_b128 then cover all 64 banks exactly once) they are 8
//     ds_read_b128 (4 cycles per 16 bytes/lane) instead of 32 ds_read_b32 (2 cycles per 4 bytes/lane): half the cycles;
//   * twiddle / ring reads: ds_read_b64 issued from inline asm in chunks of 8 with counted lgkmcnt waits, because the
//     compiler pairs its own 8-byte LDS loads into ds_read2(st64)_b64, which the LDS serves at half the bytes per clock
//     of two ds_read_b64 (a volatile load is no way out: it becomes a flat load with a full wait behind each).
typedef float v2f __attribute__((ext_vector_type(2)));
constexpr int kWgScrStride = 68;
constexpr int kWgScrFloats = 32 * kWgScrStride;

__device__ __forceinline__ unsigned lds_addr(const void* p) {
    return (unsigned)(size_t)(__attribute__((address_space(3))) const char*)p;
}
// One ds_read_b64 the compiler does not see as a memory operation (so it cannot pair it); the destination is only
// valid after a lds_wait8<N>() naming it.  LDS returns in issue order, so lgkmcnt(N) completes everything but the
// youngest N DS operations of this wave, whoever issued them.
template <int OFF>
__device__ __forceinline__ void lds_rd8(v2f& dst, unsigned addr) {
    asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "i"(OFF));
}
template <int N>
__device__ __forceinline__ void lds_wait8(v2f (&a)[8]) {
    asm volatile("s_waitcnt lgkmcnt(%8)"
                 : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7])
                 : "i"(N));
}
// Eight-at-a-time streaming of 32 table entries: body(k, value) for k = 0..31, entry k read from addr + OFF(k).
template <typename OffFn, typename Body>
__device__ __forceinline__ void lds_stream32(unsigned addr, OffFn, Body body) {
    v2f buf[2][8];
    auto issue = [&](auto cc) {
        constexpr int c = decltype(cc)::value;
        lds_rd8<OffFn::off(8 * c + 0)>(buf[c & 1][0], addr); lds_rd8<OffFn::off(8 * c + 1)>(buf[c & 1][1], addr);
        lds_rd8<OffFn::off(8 * c + 2)>(buf[c & 1][2], addr); lds_rd8<OffFn::off(8 * c + 3)>(buf[c & 1][3], addr);
        lds_rd8<OffFn::off(8 * c + 4)>(buf[c & 1][4], addr); lds_rd8<OffFn::off(8 * c + 5)>(buf[c & 1][5], addr);
        lds_rd8<OffFn::off(8 * c + 6)>(buf[c & 1][6], addr); lds_rd8<OffFn::off(8 * c + 7)>(buf[c & 1][7], addr);
    };
    issue(std::integral_constant<int, 0>{});
    issue(std::integral_constant<int, 1>{});
    lds_wait8<8>(buf[0]);
#pragma unroll
    for (int j = 0; j < 8; ++j) body(j, buf[0][j]);
    issue(std::integral_constant<int, 2>{});
    lds_wait8<8>(buf[1]);
#pragma unroll
    for (int j = 0; j < 8; ++j) body(8 + j, buf[1][j]);
    issue(std::integral_constant<int, 3>{});
    lds_wait8<8>(buf[0]);
#pragma unroll
    for (int j = 0; j < 8; ++j) body(16 + j, buf[0][j]);
    lds_wait8<0>(buf[1]);
#pragma unroll
    for (int j = 0; j < 8; ++j) body(24 + j, buf[1][j]);
}
struct OffTwl { static constexpr int off(int i) { return 512 * brev5(i); } };      // twl[brev5(i)][lane], register order
struct OffTwh { static constexpr int off(int j) { return 16 * (j & 15) + 8 * (j >> 4); } };   // twp[h][j & 15][j >> 4] from &twp[h][0][0]
struct OffRow { static constexpr int off(int k) { return 512 * (k & 15); } };      // 16 consecutive 64-entry rows
#ifndef LEAF_FFT_TWL_PAIRS
#define LEAF_FFT_TWL_PAIRS 0       // 1: first-stage twiddles in register-pair order, 16 ds_read_b128 instead of 31 ds_read_b64 --
#endif                             // measured slower (0.2188 vs 0.2178 ms whole forward, interleaved): kept as an A/B switch

// scr[brev5(i) * 68 + lane] = v[i], i = 0..31, through M0-relative add-tid stores (M0 saved and restored: the compiler
// owns it for the LDS-DMA builtin).  scr_lds = the wave's scr as an LDS byte address (wave-uniform).
__device__ __forceinline__ void wg_transpose_store(const float (&v)[32], unsigned scr_lds) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %17\n\ts_nop 0\n\t"
                 "ds_write_addtid_b32 %1 offset:0\n\t"
                 "ds_write_addtid_b32 %2 offset:4352\n\t"
                 "ds_write_addtid_b32 %3 offset:2176\n\t"
                 "ds_write_addtid_b32 %4 offset:6528\n\t"
                 "ds_write_addtid_b32 %5 offset:1088\n\t"
                 "ds_write_addtid_b32 %6 offset:5440\n\t"
                 "ds_write_addtid_b32 %7 offset:3264\n\t"
                 "ds_write_addtid_b32 %8 offset:7616\n\t"
                 "ds_write_addtid_b32 %9 offset:544\n\t"
                 "ds_write_addtid_b32 %10 offset:4896\n\t"
                 "ds_write_addtid_b32 %11 offset:2720\n\t"
                 "ds_write_addtid_b32 %12 offset:7072\n\t"
                 "ds_write_addtid_b32 %13 offset:1632\n\t"
                 "ds_write_addtid_b32 %14 offset:5984\n\t"
                 "ds_write_addtid_b32 %15 offset:3808\n\t"
                 "ds_write_addtid_b32 %16 offset:8160\n\t"
                 "s_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(v[0]), "v"(v[1]), "v"(v[2]), "v"(v[3]), "v"(v[4]), "v"(v[5]), "v"(v[6]), "v"(v[7]), "v"(v[8]), "v"(v[9]), "v"(v[10]), "v"(v[11]), "v"(v[12]), "v"(v[13]), "v"(v[14]), "v"(v[15]), "s"(scr_lds)
                 : "memory");
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %17\n\ts_nop 0\n\t"
                 "ds_write_addtid_b32 %1 offset:272\n\t"
                 "ds_write_addtid_b32 %2 offset:4624\n\t"
                 "ds_write_addtid_b32 %3 offset:2448\n\t"
                 "ds_write_addtid_b32 %4 offset:6800\n\t"
                 "ds_write_addtid_b32 %5 offset:1360\n\t"
                 "ds_write_addtid_b32 %6 offset:5712\n\t"
                 "ds_write_addtid_b32 %7 offset:3536\n\t"
                 "ds_write_addtid_b32 %8 offset:7888\n\t"
                 "ds_write_addtid_b32 %9 offset:816\n\t"
                 "ds_write_addtid_b32 %10 offset:5168\n\t"
                 "ds_write_addtid_b32 %11 offset:2992\n\t"
                 "ds_write_addtid_b32 %12 offset:7344\n\t"
                 "ds_write_addtid_b32 %13 offset:1904\n\t"
                 "ds_write_addtid_b32 %14 offset:6256\n\t"
                 "ds_write_addtid_b32 %15 offset:4080\n\t"
                 "ds_write_addtid_b32 %16 offset:8432\n\t"
                 "s_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(v[16]), "v"(v[17]), "v"(v[18]), "v"(v[19]), "v"(v[20]), "v"(v[21]), "v"(v[22]), "v"(v[23]), "v"(v[24]), "v"(v[25]), "v"(v[26]), "v"(v[27]), "v"(v[28]), "v"(v[29]), "v"(v[30]), "v"(v[31]), "s"(scr_lds)
                 : "memory");
}

// Half-buffer transposition (16-wave workgroups: the LDS only has room for 16 rows of scr per wave).  Step S stores the
// 16 registers whose row brev5(i) lies in [16 S, 16 S + 16) as rows 0..15; the lanes whose row k1r = lane & 31 lies in that
// range then read their 32 values (8 ds_read_b128 under an exec mask set inside the statement, so that the two steps
// fill the same destination registers without compiler-made copies).
template <int STEP>
__device__ __forceinline__ void wg_transpose_store_half(const float (&v)[32], unsigned scr_lds) {
    unsigned keep;
    if constexpr (STEP == 0) {
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %17\n\ts_nop 0\n\t"
                     "ds_write_addtid_b32 %1 offset:0\n\t"
                     "ds_write_addtid_b32 %2 offset:2176\n\t"
                     "ds_write_addtid_b32 %3 offset:1088\n\t"
                     "ds_write_addtid_b32 %4 offset:3264\n\t"
                     "ds_write_addtid_b32 %5 offset:544\n\t"
                     "ds_write_addtid_b32 %6 offset:2720\n\t"
                     "ds_write_addtid_b32 %7 offset:1632\n\t"
                     "ds_write_addtid_b32 %8 offset:3808\n\t"
                     "ds_write_addtid_b32 %9 offset:272\n\t"
                     "ds_write_addtid_b32 %10 offset:2448\n\t"
                     "ds_write_addtid_b32 %11 offset:1360\n\t"
                     "ds_write_addtid_b32 %12 offset:3536\n\t"
                     "ds_write_addtid_b32 %13 offset:816\n\t"
                     "ds_write_addtid_b32 %14 offset:2992\n\t"
                     "ds_write_addtid_b32 %15 offset:1904\n\t"
                     "ds_write_addtid_b32 %16 offset:4080\n\t"
                     "s_mov_b32 m0, %0"
                     : "=&s"(keep)
                     : "v"(v[0]), "v"(v[2]), "v"(v[4]), "v"(v[6]), "v"(v[8]), "v"(v[10]), "v"(v[12]), "v"(v[14]), "v"(v[16]), "v"(v[18]), "v"(v[20]), "v"(v[22]), "v"(v[24]), "v"(v[26]), "v"(v[28]), "v"(v[30]), "s"(scr_lds)
                     : "memory");
    } else {
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %17\n\ts_nop 0\n\t"
                     "ds_write_addtid_b32 %1 offset:0\n\t"
                     "ds_write_addtid_b32 %2 offset:2176\n\t"
                     "ds_write_addtid_b32 %3 offset:1088\n\t"
                     "ds_write_addtid_b32 %4 offset:3264\n\t"
                     "ds_write_addtid_b32 %5 offset:544\n\t"
                     "ds_write_addtid_b32 %6 offset:2720\n\t"
                     "ds_write_addtid_b32 %7 offset:1632\n\t"
                     "ds_write_addtid_b32 %8 offset:3808\n\t"
                     "ds_write_addtid_b32 %9 offset:272\n\t"
                     "ds_write_addtid_b32 %10 offset:2448\n\t"
                     "ds_write_addtid_b32 %11 offset:1360\n\t"
                     "ds_write_addtid_b32 %12 offset:3536\n\t"
                     "ds_write_addtid_b32 %13 offset:816\n\t"
                     "ds_write_addtid_b32 %14 offset:2992\n\t"
                     "ds_write_addtid_b32 %15 offset:1904\n\t"
                     "ds_write_addtid_b32 %16 offset:4080\n\t"
                     "s_mov_b32 m0, %0"
                     : "=&s"(keep)
                     : "v"(v[1]), "v"(v[3]), "v"(v[5]), "v"(v[7]), "v"(v[9]), "v"(v[11]), "v"(v[13]), "v"(v[15]), "v"(v[17]), "v"(v[19]), "v"(v[21]), "v"(v[23]), "v"(v[25]), "v"(v[27]), "v"(v[29]), "v"(v[31]), "s"(scr_lds)
                     : "memory");
    }
}
// t[q] = row[4 q .. 4 q + 3] for the lanes of `mask`; valid after a wait naming t (lds_wait_b128x16)
__device__ __forceinline__ void wg_transpose_load_masked(f32x4 (&t)[8], unsigned row_addr, unsigned long long mask) {
    unsigned long long save;
    asm volatile("s_mov_b64 %8, exec\n\ts_mov_b64 exec, %10\n\t"
                 "ds_read_b128 %0, %9 offset:0\n\tds_read_b128 %1, %9 offset:16\n\t"
                 "ds_read_b128 %2, %9 offset:32\n\tds_read_b128 %3, %9 offset:48\n\t"
                 "ds_read_b128 %4, %9 offset:64\n\tds_read_b128 %5, %9 offset:80\n\t"
                 "ds_read_b128 %6, %9 offset:96\n\tds_read_b128 %7, %9 offset:112\n\t"
                 "s_mov_b64 exec, %8"
                 : "+v"(t[0]), "+v"(t[1]), "+v"(t[2]), "+v"(t[3]), "+v"(t[4]), "+v"(t[5]), "+v"(t[6]), "+v"(t[7]), "=&s"(save)
                 : "v"(row_addr), "s"(mask)
                 : "memory");
}
// The first of the two masked loads of a plane: destinations are pure outputs (no value carried in), so that the compiler
// does not have to materialise -- and keep alive through the whole previous phase -- 64 registers of zeros for them.
__device__ __forceinline__ void wg_transpose_load_masked_first(f32x4 (&t)[8], unsigned row_addr, unsigned long long mask) {
    unsigned long long save;
    asm volatile("s_mov_b64 %8, exec\n\ts_mov_b64 exec, %10\n\t"
                 "ds_read_b128 %0, %9 offset:0\n\tds_read_b128 %1, %9 offset:16\n\t"
                 "ds_read_b128 %2, %9 offset:32\n\tds_read_b128 %3, %9 offset:48\n\t"
                 "ds_read_b128 %4, %9 offset:64\n\tds_read_b128 %5, %9 offset:80\n\t"
                 "ds_read_b128 %6, %9 offset:96\n\tds_read_b128 %7, %9 offset:112\n\t"
                 "s_mov_b64 exec, %8"
                 : "=&v"(t[0]), "=&v"(t[1]), "=&v"(t[2]), "=&v"(t[3]), "=&v"(t[4]), "=&v"(t[5]), "=&v"(t[6]), "=&v"(t[7]), "=&s"(save)
                 : "v"(row_addr), "s"(mask)
                 : "memory");
}
__device__ __forceinline__ void lds_wait_b128x16(f32x4 (&a)[8], f32x4 (&b)[8]) {
    asm volatile("s_waitcnt lgkmcnt(0)"
                 : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]),
                   "+v"(b[0]), "+v"(b[1]), "+v"(b[2]), "+v"(b[3]), "+v"(b[4]), "+v"(b[5]), "+v"(b[6]), "+v"(b[7]));
}

// Column-half transposition (swap-free cross stage, below): the 32 lanes of one half-wave store their 32 registers as
// rows brev5(i) of a [32][36] scratch (columns = their 32 lane positions); every lane of the wave then reads row
// lane & 31 -- once after the lower half-wave's stores, once after the upper's.  M0-relative add-tid stores under an exec
// mask set inside the statement; the upper half-wave's base is 128 bytes lower so that lane 32 lands in column 0.
constexpr int kWgColStride = 36;
constexpr int kWgScrHalfFloats = 32 * kWgColStride;                      // 4608 bytes per wave
template <int UPPER>
__device__ __forceinline__ void wg_transpose_store_cols(const float (&v)[32], unsigned scr_lds) {
    unsigned keep;
    unsigned long long save;
    const unsigned base = scr_lds - (UPPER ? 128u : 0u);
    const unsigned long long mask = UPPER ? 0xFFFFFFFF00000000ull : 0x00000000FFFFFFFFull;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b64 %1, exec\n\ts_mov_b32 m0, %18\n\ts_mov_b64 exec, %19\n\t"
                 "ds_write_addtid_b32 %2 offset:0\n\t"
                 "ds_write_addtid_b32 %3 offset:2304\n\t"
                 "ds_write_addtid_b32 %4 offset:1152\n\t"
                 "ds_write_addtid_b32 %5 offset:3456\n\t"
                 "ds_write_addtid_b32 %6 offset:576\n\t"
                 "ds_write_addtid_b32 %7 offset:2880\n\t"
                 "ds_write_addtid_b32 %8 offset:1728\n\t"
                 "ds_write_addtid_b32 %9 offset:4032\n\t"
                 "ds_write_addtid_b32 %10 offset:288\n\t"
                 "ds_write_addtid_b32 %11 offset:2592\n\t"
                 "ds_write_addtid_b32 %12 offset:1440\n\t"
                 "ds_write_addtid_b32 %13 offset:3744\n\t"
                 "ds_write_addtid_b32 %14 offset:864\n\t"
                 "ds_write_addtid_b32 %15 offset:3168\n\t"
                 "ds_write_addtid_b32 %16 offset:2016\n\t"
                 "ds_write_addtid_b32 %17 offset:4320\n\t"
                 "s_mov_b64 exec, %1\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep), "=&s"(save)
                 : "v"(v[0]), "v"(v[1]), "v"(v[2]), "v"(v[3]), "v"(v[4]), "v"(v[5]), "v"(v[6]), "v"(v[7]), "v"(v[8]), "v"(v[9]), "v"(v[10]), "v"(v[11]), "v"(v[12]), "v"(v[13]), "v"(v[14]), "v"(v[15]), "s"(base), "s"(mask)
                 : "memory");
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b64 %1, exec\n\ts_mov_b32 m0, %18\n\ts_mov_b64 exec, %19\n\t"
                 "ds_write_addtid_b32 %2 offset:144\n\t"
                 "ds_write_addtid_b32 %3 offset:2448\n\t"
                 "ds_write_addtid_b32 %4 offset:1296\n\t"
                 "ds_write_addtid_b32 %5 offset:3600\n\t"
                 "ds_write_addtid_b32 %6 offset:720\n\t"
                 "ds_write_addtid_b32 %7 offset:3024\n\t"
                 "ds_write_addtid_b32 %8 offset:1872\n\t"
                 "ds_write_addtid_b32 %9 offset:4176\n\t"
                 "ds_write_addtid_b32 %10 offset:432\n\t"
                 "ds_write_addtid_b32 %11 offset:2736\n\t"
                 "ds_write_addtid_b32 %12 offset:1584\n\t"
                 "ds_write_addtid_b32 %13 offset:3888\n\t"
                 "ds_write_addtid_b32 %14 offset:1008\n\t"
                 "ds_write_addtid_b32 %15 offset:3312\n\t"
                 "ds_write_addtid_b32 %16 offset:2160\n\t"
                 "ds_write_addtid_b32 %17 offset:4464\n\t"
                 "s_mov_b64 exec, %1\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep), "=&s"(save)
                 : "v"(v[16]), "v"(v[17]), "v"(v[18]), "v"(v[19]), "v"(v[20]), "v"(v[21]), "v"(v[22]), "v"(v[23]), "v"(v[24]), "v"(v[25]), "v"(v[26]), "v"(v[27]), "v"(v[28]), "v"(v[29]), "v"(v[30]), "v"(v[31]), "s"(base), "s"(mask)
                 : "memory");
}

// Twiddle tables of the workgroup kernels: twl as fft_build_twiddles (LEAF_FFT_TWL_PAIRS = 1: in register-pair order,
// twq[pr][l][w] = W_2048^(l brev5(2 pr + w)), one ds_read_b128 per two registers -- measured slower); the half-wave twiddles in
// PAIR order,
//   twp[h][j][w]  = h ? W_64^(j + 16 w) : 1,  j < 16, w < 2   (float2; same 64 entries as twh[j][h]),
// so that the fused first stage of the second 32-point transform gets both twiddles of a register pair (j, j + 16) with one
// ds_read_b128 (16 LDS instructions per transform instead of 32).
__device__ __forceinline__ void fft_build_twiddles_wg(float2* twl, float2* twp, int tid, int nthreads) {
    for (int i = tid; i < 32 * 64; i += nthreads) {
#if LEAF_FFT_TWL_PAIRS
        const int pr = i >> 7, l = (i >> 1) & 63, k1 = brev5(2 * pr + (i & 1));   // twq[pr][l][w] = W_2048^(l brev5(2 pr + w))
#else
        const int k1 = i >> 6, l = i & 63;                                       // twl[k1][l]
#endif
        float s, c;
        sincospif(2.0f * (float)((l * k1) & (kFftN - 1)) / (float)kFftN, &s, &c);
        twl[i] = make_float2(c, -s);
    }
    for (int i = tid; i < 64; i += nthreads) {
        const int h = i >> 5, e = i & 31, j = (e >> 1) + 16 * (e & 1);
        float s, c;
        sincospif(2.0f * (float)j / 64.0f, &s, &c);
        twp[i] = h ? make_float2(c, -s) : make_float2(1.0f, 0.0f);
    }
}
// sixteen 16-byte table entries at addr + STRIDE k, four at a time, two groups in flight: body(k, value)
template <int OFF>
__device__ __forceinline__ void lds_rd16(f32x4& dst, unsigned addr) {
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "i"(OFF));
}
template <int N>
__device__ __forceinline__ void lds_wait16x4(f32x4 (&a)[4]) {
    asm volatile("s_waitcnt lgkmcnt(%4)" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]) : "i"(N));
}
template <int STRIDE = 16, typename Body>
__device__ __forceinline__ void lds_stream16q(unsigned addr, Body body) {
    f32x4 buf[2][4];
    auto issue = [&](auto cc) {
        constexpr int c = decltype(cc)::value;
        lds_rd16<STRIDE * (4 * c + 0)>(buf[c & 1][0], addr); lds_rd16<STRIDE * (4 * c + 1)>(buf[c & 1][1], addr);
        lds_rd16<STRIDE * (4 * c + 2)>(buf[c & 1][2], addr); lds_rd16<STRIDE * (4 * c + 3)>(buf[c & 1][3], addr);
    };
    issue(std::integral_constant<int, 0>{});
    issue(std::integral_constant<int, 1>{});
    lds_wait16x4<4>(buf[0]);
#pragma unroll
    for (int j = 0; j < 4; ++j) body(j, buf[0][j]);
    issue(std::integral_constant<int, 2>{});
    lds_wait16x4<4>(buf[1]);
#pragma unroll
    for (int j = 0; j < 4; ++j) body(4 + j, buf[1][j]);
    issue(std::integral_constant<int, 3>{});
    lds_wait16x4<4>(buf[0]);
#pragma unroll
    for (int j = 0; j < 4; ++j) body(8 + j, buf[0][j]);
    lds_wait16x4<0>(buf[1]);
#pragma unroll
    for (int j = 0; j < 4; ++j) body(12 + j, buf[1][j]);
}

}  // namespace
#include "leaf_band.hpp"           // band-limited filter tasks (uses the LDS helpers above)
namespace {

#ifndef LEAF_FFT_FUSE_TWIDDLE
#define LEAF_FFT_FUSE_TWIDDLE 1    // 0: separate half-wave twiddle products, then the full 32-point transform (A/B)
#endif
#ifndef LEAF_FFT_NOSWAP
#define LEAF_FFT_NOSWAP 1          // 0: the v_permlane32_swap exchange of round 2's first kernels (A/B measurements)
#endif
// SKIP1: the caller has already run the first decimation-in-time stage (registers (k, k + 16), unit twiddles) -- fused with the
// spectral multiply that produced the input (wg_multiply_stage1)
template <bool HALF, bool SKIP1 = false>
__device__ __forceinline__ void fft2048w(float (&re)[32], float (&im)[32], float* scr, unsigned scr_lds, const float2* twl,
                                         const float2* twh, int lane) {
    if constexpr (SKIP1) {
        static_assert(LEAF_FFT32_DIT, "the fused first stage is the decimation-in-time one");
        fft32_dit_stage<2>(re, im);
        fft32_dit_stage<4>(re, im);
        fft32_dit_stage<8>(re, im);
        fft32_dit_stage<16>(re, im);
    } else {
        fft32_dif(re, im);                               // register i <-> k1 = brev5(i), lane = n2
    }
#if LEAF_FFT_TWL_PAIRS
    lds_stream16q<1024>(lds_addr(twl + 2 * lane), [&](int pr, f32x4 w) {    // (w.x, w.y), (w.z, w.w): twiddles of registers 2 pr, 2 pr + 1
        if (pr > 0) {                                     // register 0: W^0 = 1
            const float r = re[2 * pr] * w.x - im[2 * pr] * w.y;
            im[2 * pr] = re[2 * pr] * w.y + im[2 * pr] * w.x;
            re[2 * pr] = r;
        }
        const float r1 = re[2 * pr + 1] * w.z - im[2 * pr + 1] * w.w;
        im[2 * pr + 1] = re[2 * pr + 1] * w.w + im[2 * pr + 1] * w.z;
        re[2 * pr + 1] = r1;
    });
#else
    lds_stream32(lds_addr(twl + lane), OffTwl{}, [&](int i, v2f w) {
        if (i == 0) return;                               // W^0 = 1
        const float r = re[i] * w.x - im[i] * w.y;
        im[i] = re[i] * w.y + im[i] * w.x;
        re[i] = r;
    });
#endif
    const int k1r = lane & 31, h = lane >> 5;
    float tr[32], ti[32];
#if LEAF_FFT_NOSWAP
    // 64-point DFT over n2 = j + 32 hh, first radix-2 step WITHOUT a cross-lane exchange: every lane reads both halves of
    // its transposed row (its own 32 columns and the other half-wave's) and forms a + b (lower half-wave) or a - b (upper)
    // itself -- one FMA per value.  v_permlane32_swap issues at ~8 cycles (profiles/r01/ubench_valu.txt), the 64 swaps
    // per transform of the exchange form were a sixth of its issue time; the price is 16 more ds_read_b128 per transform.
    const float sg = h ? -1.0f : 1.0f;                    // t = a + sg b: a = x[j] (lower half-wave's), b = x[j + 32] (upper's)
    if constexpr (HALF) {
        const f32x4* row = reinterpret_cast<const f32x4*>(scr + k1r * kWgColStride);
        auto plane = [&](const float (&src)[32], float (&t)[32]) {
            f32x4 a[8], b[8];
            wg_transpose_store_cols<0>(src, scr_lds);
#pragma unroll
            for (int q = 0; q < 8; ++q) a[q] = row[q];
            wg_transpose_store_cols<1>(src, scr_lds);     // LDS executes in order: the reads above are served first
#pragma unroll
            for (int q = 0; q < 8; ++q) b[q] = row[q];
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                t[4 * q] = fmaf(b[q].x, sg, a[q].x); t[4 * q + 1] = fmaf(b[q].y, sg, a[q].y);
                t[4 * q + 2] = fmaf(b[q].z, sg, a[q].z); t[4 * q + 3] = fmaf(b[q].w, sg, a[q].w);
            }
            asm volatile("" ::: "memory");
        };
        pin32(re);                                        // the twiddle products are complete before the first store
        pin32(im);
        plane(re, tr);
        pin32(tr);
        plane(im, ti);
        pin32(ti);
    } else {
        const f32x4* lo = reinterpret_cast<const f32x4*>(scr + k1r * kWgScrStride);
        auto plane = [&](const float (&src)[32], float (&t)[32]) {
            wg_transpose_store(src, scr_lds);
#pragma unroll
            for (int q0 = 0; q0 < 8; q0 += 4) {           // 32 values in flight at a time
#pragma unroll
                for (int q = q0; q < q0 + 4; ++q) {
                    const f32x4 a = lo[q], b = lo[q + 8];
                    t[4 * q] = fmaf(b.x, sg, a.x); t[4 * q + 1] = fmaf(b.y, sg, a.y);
                    t[4 * q + 2] = fmaf(b.z, sg, a.z); t[4 * q + 3] = fmaf(b.w, sg, a.w);
                }
                asm volatile("" ::: "memory");
            }
        };
        pin32(re);
        pin32(im);
        plane(re, tr);
        pin32(tr);
        plane(im, ti);
        pin32(ti);
    }
#else
    if constexpr (HALF) {
        const unsigned row_addr = scr_lds + 4 * ((k1r & 15) * kWgScrStride + 32 * h);
        constexpr unsigned long long kLo = 0x0000FFFF0000FFFFull, kHi = 0xFFFF0000FFFF0000ull;   // lanes with k1r < 16 / >= 16
        f32x4 qr[8], qi[8];
        wg_transpose_store_half<0>(re, scr_lds);
        wg_transpose_load_masked_first(qr, row_addr, kLo);
        wg_transpose_store_half<1>(re, scr_lds);
        wg_transpose_load_masked(qr, row_addr, kHi);
        wg_transpose_store_half<0>(im, scr_lds);
        wg_transpose_load_masked_first(qi, row_addr, kLo);
        wg_transpose_store_half<1>(im, scr_lds);
        wg_transpose_load_masked(qi, row_addr, kHi);
        lds_wait_b128x16(qr, qi);
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            tr[4 * q] = qr[q].x; tr[4 * q + 1] = qr[q].y; tr[4 * q + 2] = qr[q].z; tr[4 * q + 3] = qr[q].w;
            ti[4 * q] = qi[q].x; ti[4 * q + 1] = qi[q].y; ti[4 * q + 2] = qi[q].z; ti[4 * q + 3] = qi[q].w;
        }
    } else {
        const f32x4* row = reinterpret_cast<const f32x4*>(scr + k1r * kWgScrStride + 32 * h);
        wg_transpose_store(re, scr_lds);
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const f32x4 t = row[q];
            tr[4 * q] = t.x; tr[4 * q + 1] = t.y; tr[4 * q + 2] = t.z; tr[4 * q + 3] = t.w;
        }
        asm volatile("" ::: "memory");
        wg_transpose_store(im, scr_lds);
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const f32x4 t = row[q];
            ti[4 * q] = t.x; ti[4 * q + 1] = t.y; ti[4 * q + 2] = t.z; ti[4 * q + 3] = t.w;
        }
    }
    auto cross = [](float& x0, float& x1) {
        auto g = __builtin_amdgcn_permlane32_swap(__float_as_uint(x0), __float_as_uint(x1), false, false);
        const float a = __uint_as_float(g[0]), b = __uint_as_float(g[1]);
        auto qq = __builtin_amdgcn_permlane32_swap(__float_as_uint(a + b), __float_as_uint(a - b), false, false);
        x0 = __uint_as_float(qq[0]);
        x1 = __uint_as_float(qq[1]);
    };
#pragma unroll
    for (int j = 0; j < 32; j += 2) {
        cross(tr[j], tr[j + 1]);
        cross(ti[j], ti[j + 1]);
    }
#endif
#if LEAF_FFT32_DIT && LEAF_FFT_FUSE_TWIDDLE
    // The half-wave twiddle fused into the first decimation-in-time stage of the 32-point transform over j: that stage
    // pairs registers (j, j + 16) with unit twiddles, so with a = t[j] w[j] and b = t[j + 16] w[j + 16]
    //     out[j] = a + b = a + w_b t_b  (four FMAs on top of a),   out[j + 16] = a - b = 2 a - out[j]  (two FMAs):
    // 10 instructions per pair instead of 8 (two complex products) + 4 (the butterfly) -- 32 fewer per transform.  The table
    // is streamed in pair order (w[j], w[j + 16]).
    lds_stream16q(lds_addr(twh + 32 * h), [&](int j, f32x4 w) {             // w = (w[j], w[j + 16]) of this half-wave
        float ar, ai;
        if (j == 0) {
            ar = tr[0];
            ai = ti[0];
        } else {
            ar = tr[j] * w.x - ti[j] * w.y;
            ai = tr[j] * w.y + ti[j] * w.x;
        }
        const float br = tr[j + 16], bi = ti[j + 16];
        const float pr = fmaf(-bi, w.w, fmaf(br, w.z, ar));
        const float pi = fmaf(bi, w.z, fmaf(br, w.w, ai));
        re[j] = pr;
        im[j] = pi;
        re[j + 16] = fmaf(2.0f, ar, -pr);
        im[j + 16] = fmaf(2.0f, ai, -pi);
    });
    fft32_dit_stage<2>(re, im);
    fft32_dit_stage<4>(re, im);
    fft32_dit_stage<8>(re, im);
    fft32_dit_stage<16>(re, im);                         // register i <-> k' = brev5(i): element 64 k' + lane
#else
    lds_stream32(lds_addr(twh + 32 * h), OffTwh{}, [&](int j, v2f w) {
        if (j == 0) {
            re[j] = tr[j];
            im[j] = ti[j];
        } else {
            re[j] = tr[j] * w.x - ti[j] * w.y;
            im[j] = tr[j] * w.y + ti[j] * w.x;
        }
    });
    fft32_dif(re, im);                                   // register i <-> k' = brev5(i): element 64 k' + lane
#endif
}

// ---- packed front half (LEAF_WG_PK, round 4 experiment -> see DESIGN.md section 9) -------------------------------------------
// At three waves per SIMD the scarce resource is VALU issue slots, and a packed fp32 instruction spends one slot on two flops per
// lane (tools/ubench_valu.hip: the 32-point transform on complex register pairs takes 942 SIMD cycles against 1 160 in scalar
// form).  Everything of a filter task BEFORE the LDS transposition can run on complex pairs without a single shuffle, because
// those values are computed into registers of our choosing: the spectral multiply (its operands arrive from LDS as (re, im)
// pairs already), the first 32-point transform and the first-level twiddle products.  op_sel picks the halves of each source
// (swapped operands), neg_lo / neg_hi flip signs: a complex product is two packed instructions, a twiddled butterfly three.
// After the transposition the data comes back plane by plane (ds_read_b128 into consecutive registers) and stays scalar.
__device__ __forceinline__ void pk_add(v2f& d, const v2f& a, const v2f& b) { asm volatile("v_pk_add_f32 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b)); }
__device__ __forceinline__ void pk_sub(v2f& d, const v2f& a, const v2f& b) {
    asm volatile("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(d) : "v"(a), "v"(b));
}
// a + (-i) b = (a.x + b.y, a.y - b.x);  a - (-i) b = (a.x - b.y, a.y + b.x)
__device__ __forceinline__ void pk_add_mib(v2f& d, const v2f& a, const v2f& b) {
    asm volatile("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_hi:[0,1]" : "=v"(d) : "v"(a), "v"(b));
}
__device__ __forceinline__ void pk_sub_mib(v2f& d, const v2f& a, const v2f& b) {
    asm volatile("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1]" : "=v"(d) : "v"(a), "v"(b));
}
// the DIT butterfly a' = a + w b, b' = 2 a - a' with w = C + i S held as the register pair (C, S):
//   ROT = false: w = (C, S);   ROT = true: w = -i (C, S) = (S, -C)  (W_32^(8 + k) from the pair of W_32^k)
template <bool ROT>
__device__ __forceinline__ void pk_bfly(v2f& a, v2f& b, const v2f& w, const v2f& two) {
    v2f t;
    if constexpr (!ROT) {
        asm volatile("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,0,0] op_sel_hi:[1,0,1]" : "=v"(t) : "v"(b), "v"(w), "v"(a));             // a + b (C, C)
        asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,1,0] op_sel_hi:[0,1,1] neg_lo:[0,1,0]" : "+v"(t) : "v"(b), "v"(w));      // + (b.y, b.x) (-S, S)
    } else {
        asm volatile("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,1,0] op_sel_hi:[1,1,1]" : "=v"(t) : "v"(b), "v"(w), "v"(a));             // a + b (S, S)
        asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,0,0] op_sel_hi:[0,0,1] neg_hi:[0,1,0]" : "+v"(t) : "v"(b), "v"(w));      // + (b.y, b.x) (C, -C)
    }
    asm volatile("v_pk_fma_f32 %0, %1, %2, %3 neg_lo:[0,0,1] neg_hi:[0,0,1]" : "=v"(b) : "v"(two), "v"(a), "v"(t));                   // 2 a - a'
    a = t;
}
// one decimation-in-time stage of the 32-point transform on complex pairs (the register conventions of fft32_dit_stage)
template <int HALF>
__device__ __forceinline__ void pk_dit_stage(v2f (&z)[32], const v2f (&W)[8], const v2f& two) {
#pragma unroll
    for (int blk = 0; blk < 32; blk += 2 * HALF) {
#pragma unroll
        for (int j = 0; j < HALF; ++j) {
            const int a = brev5(blk + j), b = brev5(blk + j + HALF);
            constexpr int STEP = 16 / HALF;
            const int tw = j * STEP;                                        // w = W_32^tw
            if (tw == 0) {
                v2f s0;
                pk_add(s0, z[a], z[b]);
                pk_sub(z[b], z[a], z[b]);
                z[a] = s0;
            } else if (tw == 8) {
                v2f s0;
                pk_add_mib(s0, z[a], z[b]);
                pk_sub_mib(z[b], z[a], z[b]);
                z[a] = s0;
            } else if (tw < 8) {
                pk_bfly<false>(z[a], z[b], W[tw], two);
            } else {
                pk_bfly<true>(z[a], z[b], W[tw - 8], two);
            }
        }
    }
}
// the twiddle constants of the 32-point transform as register pairs: W[k] = (cos, -sin)(2 pi k / 32), k = 1..7 (W[0] unused)
__device__ __forceinline__ void pk_twiddle_pairs(v2f (&W)[8], v2f& two) {
    constexpr float C[8] = {1.0f, 0.98078528f, 0.923879533f, 0.831469612f, 0.707106781f, 0.555570233f, 0.382683432f, 0.195090322f};
    constexpr float S[8] = {0.0f, -0.195090322f, -0.382683432f, -0.555570233f, -0.707106781f, -0.831469612f, -0.923879533f, -0.98078528f};
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        W[k] = v2f{C[k], S[k]};
        asm volatile("" : "+v"(W[k]));                                    // live in registers, not re-materialised literal by literal
    }
    two = v2f{2.0f, 2.0f};
    asm volatile("" : "+v"(two));
}
// scr[brev5(i) * 68 + lane] = z[i].x (COMP = 0) or .y (COMP = 1): wg_transpose_store on one component of the pairs
template <int COMP>
__device__ __forceinline__ void wg_transpose_store_pk(const v2f (&z)[32], unsigned scr_lds) {
    float v[32];
#pragma unroll
    for (int i = 0; i < 32; ++i) v[i] = COMP ? z[i].y : z[i].x;            // sub-register views: no instruction
    wg_transpose_store(v, scr_lds);
}
// fft2048w<false, SKIP1 = true> with the front half on complex pairs: z holds the output of the fused first stage
__device__ __forceinline__ void fft2048w_pkfront(v2f (&z)[32], float (&re)[32], float (&im)[32], float* scr, unsigned scr_lds,
                                                 const float2* twl, const float2* twh, int lane, const v2f (&W)[8], const v2f& two) {
    pk_dit_stage<2>(z, W, two);
    pk_dit_stage<4>(z, W, two);
    pk_dit_stage<8>(z, W, two);
    pk_dit_stage<16>(z, W, two);                                          // register i <-> k1 = brev5(i), lane = n2
    lds_stream32(lds_addr(twl + lane), OffTwl{}, [&](int i, v2f w) {
        if (i == 0) return;                                               // W^0 = 1
        v2f t;
        asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel:[0,0] op_sel_hi:[1,0]" : "=v"(t) : "v"(z[i]), "v"(w));                         // z (w.x, w.x)
        asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,1,0] op_sel_hi:[0,1,1] neg_lo:[0,1,0]" : "+v"(t) : "v"(z[i]), "v"(w)); // + (z.y, z.x) (-w.y, w.y)
        z[i] = t;
    });
    const int k1r = lane & 31, h = lane >> 5;
    float tr[32], ti[32];
    const float sg = h ? -1.0f : 1.0f;
    const f32x4* lo = reinterpret_cast<const f32x4*>(scr + k1r * kWgScrStride);
    auto plane = [&](auto comp, float (&t)[32]) {
        wg_transpose_store_pk<decltype(comp)::value>(z, scr_lds);
#pragma unroll
        for (int q0 = 0; q0 < 8; q0 += 4) {
#pragma unroll
            for (int q = q0; q < q0 + 4; ++q) {
                const f32x4 a = lo[q], b = lo[q + 8];
                t[4 * q] = fmaf(b.x, sg, a.x); t[4 * q + 1] = fmaf(b.y, sg, a.y);
                t[4 * q + 2] = fmaf(b.z, sg, a.z); t[4 * q + 3] = fmaf(b.w, sg, a.w);
            }
            asm volatile("" ::: "memory");
        }
    };
#pragma unroll
    for (int i = 0; i < 32; ++i) asm volatile("" : "+v"(z[i]));          // the twiddle products are complete before the first store
    plane(std::integral_constant<int, 0>{}, tr);
    pin32(tr);
    plane(std::integral_constant<int, 1>{}, ti);
    pin32(ti);
    lds_stream16q(lds_addr(twh + 32 * h), [&](int j, f32x4 w) {             // the fused half-wave twiddle + first stage of fft2048w
        float ar, ai;
        if (j == 0) {
            ar = tr[0];
            ai = ti[0];
        } else {
            ar = tr[j] * w.x - ti[j] * w.y;
            ai = tr[j] * w.y + ti[j] * w.x;
        }
        const float br = tr[j + 16], bi = ti[j + 16];
        const float pr = fmaf(-bi, w.w, fmaf(br, w.z, ar));
        const float pi = fmaf(bi, w.z, fmaf(br, w.w, ai));
        re[j] = pr;
        im[j] = pi;
        re[j + 16] = fmaf(2.0f, ar, -pr);
        im[j + 16] = fmaf(2.0f, ai, -pi);
    });
    fft32_dit_stage<2>(re, im);
    fft32_dit_stage<4>(re, im);
    fft32_dit_stage<8>(re, im);
    fft32_dit_stage<16>(re, im);
}
#ifndef LEAF_WG_PK
#define LEAF_WG_PK 0               // 1: the static forward kernels run the front half of a filter task on packed complex pairs (A/B)
#endif

// Task ids over a DENSE grid of F + 1 slots per set: task 0 = fwd(0); task 1 + i (F + 1) + r = slot r of set i, slot 0 being
// fwd(i + 1) and slots 1..F the set's filters.  u / (F + 1) by multiply-high with M = ceil(2^32 / (F + 1)) -- exact while
// u (M (F + 1) - 2^32) < 2^32, i.e. u < 2^32 / (F + 1); a plain division beyond that (`exact` = false).  (A power-of-two grid
// decoded by shifts left 23 of 64 slots empty at F = 40 -- 63 of 128 at F = 64 -- each costing a queue pull and a wasted
// spectrum-row prefetch: tools/trace_wg.py.)
struct WgTaskGrid {
    unsigned n, magic;
    bool exact;
};
__device__ __forceinline__ WgTaskGrid wg_task_grid(int F, int nset) {
    WgTaskGrid g;
    g.n = (unsigned)F + 1u;
    g.magic = (unsigned)((0x100000000ull + g.n - 1u) / g.n);
    g.exact = (unsigned long long)(nset > 0 ? nset : 1) * g.n * g.n < 0x80000000ull;
    return g;
}
__device__ __forceinline__ void wg_task_decode(const WgTaskGrid& g, int t, int& set, int& role) {
    if (t == 0) { set = 0; role = 0; return; }
    const unsigned u = (unsigned)(t - 1);
    const unsigned qv = g.exact ? __umulhi(u, g.magic) : u / g.n;
    set = (int)qv;
    role = (int)(u - qv * g.n);
    if (role == 0) set += 1;                                              // the NEXT set's spectrum, ahead of this set's filters
}

// floats of dynamic LDS for NW waves and a static pooling row of GU floats
constexpr int fft_wg_row_floats(int SK) { return (kGPad + SK + 63 + 3) / 4 * 4; }
constexpr int fft_wg_scr_floats(int NW) { return NW > 12 ? kWgScrHalfFloats : kWgScrFloats; }   // > 12 waves: half buffer
#ifndef LEAF_WG_REGW
#define LEAF_WG_REGW 1             // static forward kernels: pooling weights in registers (0: round 2's wave-private LDS row, A/B)
#endif
constexpr size_t fft_wg_lds_bytes(int NW, int SK) {
    return ((size_t)kTwFloats + 2 * 2 * kWgRingFwdFloat2 + kWgQueueInts +
            (size_t)NW * (fft_wg_scr_floats(NW) + (LEAF_WG_REGW ? 0 : fft_wg_row_floats(SK)))) * 4;
}
// Static pooling with the weights in REGISTERS.  A 64-sample row r of |y|^2 meets frame fi's window at window index
// j0 + lane with j0 = 64 r - is(fi), is(fi) = (DMIN + fi) hop - padL: every j0 is congruent to padL modulo g = gcd(64, hop),
// so the ~80 (row, frame) pairs of a block share only NJ = (K + 62 - jmin) / g + 1 distinct weight vectors
// w_k[lane] = g_f[jmin + g k + lane] (15 at 401/160, 14 at 801/320, 17 at 201/80; zero outside the window: the table row has
// kGPad zeros in front and a zero tail).  They are read from the filter's table row once per task (NJ coalesced loads) and
// replace the wave-private LDS copy of the row, its DMA and one ds_read_b32 per pair.
constexpr int wg_gcd(int a, int b) { return b == 0 ? a : wg_gcd(b, a % b); }
constexpr int wg_pool_step(int SHOP) { return wg_gcd(64, SHOP); }
constexpr int wg_pool_jmin(int SK, int SHOP) {       // smallest j0 >= -63 congruent to padL modulo the step
    const int g = wg_pool_step(SHOP), padl = SK / 2 + SK % 2 - 1;
    return -63 + (((padl + 63) % g) + g) % g;
}
constexpr int wg_pool_nj(int SK, int SHOP) { return (SK - 1 - wg_pool_jmin(SK, SHOP)) / wg_pool_step(SHOP) + 1; }

#ifndef LEAF_WG_TAIL
#define LEAF_WG_TAIL 1             // 0: no clip-resident finalize code in the kernel (A/B; the host must then not set fin_fused)
#endif
#ifndef LEAF_WG_STRIDED
#define LEAF_WG_STRIDED 0          // 1: blocks dealt by striding (round 2) instead of contiguously (A/B)
#endif
// ---- streaming finalize (STREAM = true) -------------------------------------------------------------------------------
// When the dealing gives every workgroup whole clips, the per-frame partial sums never leave the CU: an inverse task drops
// its (at most NFR) frame sums into a small LDS ring, fr[frame mod RING][filter] (ONE accumulator per frame and filter since
// round 5: the two blocks a window meets add their sums with ds_add_f32 -- a + b either way round, the rounding of `part`'s
// slot 0 + slot 1 -- and the finalize puts the zero back; the ring is twice as long in the same LDS, which the band tasks'
// shorter blocks need: with a lag of 2 the forward task waited ~25 k cycles per 20 k-cycle block), and once every filter of
// block c is through, the frames that block completed are finalized
// by ONE wave, lane = filter: sums -> bias -> floor -> the EMA recurrence, continued from the state kept in LDS -> PCEN ->
// the output rows (fin_* of leaf_fft.hpp: the same arithmetic as the row kernel, so a clip's bits do not depend on which
// of the two finalizes it).  That wave is whichever finishes the block's LAST filter (the completion counter tells it), so
// nobody waits for stragglers; blocks are finalized in order (a done-counter per ring-slot parity), and the forward task of
// block c + 2 publishes its spectrum only after block c's frames are out, so that no inverse task of block c + 2 -- whose
// frames wrap onto them in the ring -- starts earlier.  No partial sums in HBM, no second kernel, no tail.  Frames are numbered through the workgroup's clips (clip ordinal x
// T' + m), so that the last frames of one clip and the first of the next sit side by side in the ring.
// The ring length (a power of two, FftParams::stream_ring, chosen by the host as large as the LDS allows) sets how far ahead
// the forward tasks may run: block s may publish its spectrum once block s - D is out, D = wg_stream_lag(ring) (even, so
// that both sit on the same ring-slot parity).  The smallest ring gives D = 2 -- the forward wave then idles until the
// block two behind is finalized, ~0.8 task per block, measured +5 % on the kernel; twice that ring gives D >= 4 and
// nobody waits.
constexpr int wg_pow2_ceil(int v) { int r = 1; while (r < v) r <<= 1; return r; }
// filters-done counters: one per block modulo 8 (sq[set & 7], behind the ring; target ((set >> 3) + 1) F).  Blocks 8 apart share one, so
// the forward lag is capped at 6: block s + 8 cannot start before block s + 2 is finalized, i.e. long after block s.
constexpr int kWgStreamMaxLag = 6;
constexpr int wg_stream_lag(int SK, int SHOP, int ring) {
    const int padl = SK / 2 + SK % 2 - 1, ls = fft_block_len(SK, SHOP, true);
    const int dmax = (ls - 1 + padl) / SHOP, mf = (ls - SK + padl) / SHOP, fb = ls / SHOP;
    // frames written by block s reach s fb + dmax; block j is final through j fb + mf: (s fb + dmax) - ring <= j fb + mf
    const int lag = ((ring + mf - dmax) / fb) & ~1;
    return lag > kWgStreamMaxLag ? kWgStreamMaxLag : lag;
}
constexpr int wg_stream_ring_min(int SK, int SHOP) {                      // smallest ring with a lag of 2
    int r = 8;
    while (wg_stream_lag(SK, SHOP, r) < 2) r <<= 1;
    return r;
}
constexpr int wg_stream_fp(int F) { return F | 1; }                       // filters padded to an odd count (bank spread)
// Output staging of the streaming finalize: a block completes ~10 frames of every filter's row, which written as they come
// are 4-byte stores at stride T' (measured 2.4x the output's bytes at the memory side, round 3).  The finished values are
// parked in LDS instead, [F][kWgOutRow] (two chunks of 32 frames, by chunk parity, + 1 pad), and a 32-frame chunk of every
// row goes out together once its last frame is final (or the clip ends): 128 contiguous bytes per row and store
// instruction.  A block completes fewer than 32 frames, so it touches at most two consecutive chunks, and the older of the
// two slots it may write held a chunk that an earlier block completed and flushed.
constexpr int kWgOutChunk = 32;
constexpr int kWgOutRow = 2 * kWgOutChunk + 1;
// Round 5: the finalizing wave raises its issue priority for the duration.  With the band tasks a block is ~19 tasks, and the
// blocks' finalizes are ONE dependent chain through the launch (in order: the EMA state) of ~900 latency-bound instructions per
// block on a wave that shares its SIMD with two transform waves: traced at 7-30 k cycles of a ~21 k-cycle block period, each
// starting 11-35 k cycles late -- 13-16 % of the kernel.  s_setprio 3: BASELINE configs[4] 1.0605 -> 0.8946 ms, configs[3]'s shape
// 0.2390 -> 0.2073 ms, same box (profiles/r05/ab_stream_finalize.txt; now ahead of the partial-sum path it replaces).  Also
// measured there: five frames interleaved instead of four (nothing) and every filter task finalizing its own row with lane =
// frame (8-10 % slower: bookkeeping on every task, a point function per 30 frames of one filter instead of per 400 of forty).
#ifndef LEAF_WG_SPEC0
#define LEAF_WG_SPEC0 1            // the workgroups' first-block spectra come from the table launch (0: every forward transform in the kernel, A/B)
#endif
#ifndef LEAF_STREAM_PRIO
#define LEAF_STREAM_PRIO 1         // the finalizing wave raises its issue priority (s_setprio 3) for the duration (0: A/B)
#endif
constexpr size_t fft_wg_stream_lds_bytes(int NW, int SK, int ring, int F) {      // + frame ring, EMA state, per-filter coefficients, output staging
    return fft_wg_lds_bytes(NW, SK) + ((size_t)ring * wg_stream_fp(F) + wg_stream_fp(F) + 8 * (size_t)F + 8 +
                                       (size_t)F * kWgOutRow) * 4;
}

template <int SK, int SHOP, int NW, bool STREAM = false>
__global__ __launch_bounds__(NW * 64, (NW + 3) / 4) void leaf_fft_wg_kernel(const FftParams p) {
    constexpr bool HALF = NW > 12;                                        // 16 rows of transposition scratch per wave
    constexpr int SCRF = fft_wg_scr_floats(NW);
    extern __shared__ __attribute__((aligned(16))) float wsm[];
    float2* twl = reinterpret_cast<float2*>(wsm);                        // [32][64]
    float2* twh = twl + 32 * 64;                                          // [32][2]
    float2* ring = twh + 64;                                              // [2][kWgRingFwdFloat2]
    int* q = reinterpret_cast<int*>(ring + 2 * kWgRingFwdFloat2);         // q_next | fwd_cnt[2] | inv_cnt[2]
    constexpr int GU = LEAF_WG_REGW ? 0 : fft_wg_row_floats(SK);
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane0 = tid & 63;
    float* scr = reinterpret_cast<float*>(q + kWgQueueInts) + (size_t)wave * (SCRF + GU);
    float* sG = scr + SCRF;
    (void)sG;
    // STREAM: frame ring [RING][FPS] (one accumulator per frame and filter: the blocks a window meets ADD their sums, the
    // finalize reads it and puts the zero back) and the EMA state [FPS] behind the waves' scratch
    const int RING = p.stream_ring;                                       // power of two (host: fft_forward)
    const int FPS = wg_stream_fp(p.F);
    float* fr = reinterpret_cast<float*>(q + kWgQueueInts) + (size_t)NW * (SCRF + GU);
    float* ema_st = fr + (size_t)RING * FPS;
    FinCoef* coefT = reinterpret_cast<FinCoef*>(ema_st + FPS);             // [F]: the filters' finalize coefficients, once per launch
    static_assert(sizeof(FinCoef) == 32, "8 floats per filter");
    int* sq = reinterpret_cast<int*>(coefT + p.F);                        // [8]: filters done per block (modulo 8)
    float* ost = reinterpret_cast<float*>(sq + 8);                        // [F][kWgOutRow]: finished outputs, two chunks by parity
    (void)fr; (void)ema_st; (void)coefT; (void)sq; (void)ost;
    // fin_fused == 3 (not STREAM): the per-frame sums of the clips this workgroup owns, [clip][filter][T'] floats behind the
    // waves' scratch -- every workgroup gets whole clips and they fit (cfg1: one clip, 40 x 100 x 4 B = 16 KB).  The two
    // blocks a window meets add their sums with ds_add_f32 (a + b either way round: the rounding of `part`'s slot 0 + slot 1),
    // the tail reads them from LDS: no partial sums in HBM for these clips.
    float* lsum = reinterpret_cast<float*>(q + kWgQueueInts) + (size_t)NW * (SCRF + GU);
    const bool lds_sums = !STREAM && p.fin_fused == 3;
    const unsigned scr_lds = __builtin_amdgcn_readfirstlane(
        (unsigned)(size_t)(__attribute__((address_space(3))) float*)scr);   // LDS byte address of this wave's scr
    // band-limited filter tasks (leaf_band.hpp): the plan sits behind everything else
    constexpr bool BANDK = !HALF && band_geometry_ok(SK, SHOP) && LEAF_WG_REGW && LEAF_FFT32_DIT && LEAF_FFT_FUSE_TWIDDLE && !LEAF_WG_PK &&
                           !LEAF_FFT_TWL_PAIRS;
    const bool band_on = BANDK && p.band.rec != nullptr;
    int* bl = reinterpret_cast<int*>(wsm + p.band.lds_off);
    (void)bl;
    if constexpr (BANDK) {
        if (band_on && wave == 0) band_build_plan(p.band.rec, p.band.elist, p.band.n_edge, p.F, bl, lane0, p.band.bias, p.band.smax);
    }

    // The workgroup's first block: its spectrum was computed by the table launch on otherwise idle waves (p.spec0; this transform is
    // the one task nothing here could overlap with: eleven waves waited ~12 k cycles for it).  Wave 1 requests it before the tables
    // are built and publishes it right after the barrier; the task queue then starts behind fwd(0).
    const bool spec_pre = LEAF_WG_SPEC0 && p.spec0 != nullptr && !HALF && !LEAF_WG_STRIDED && p.B * p.nblk > 0;
    float2 sv[kWgFwdBins / 64];                                           // bins 0..1151 (kWgFwdBins)
    if (spec_pre && wave == 1) {
        const float2* src = p.spec0 + (size_t)blockIdx.x * kWgRingFwdFloat2;
#pragma unroll
        for (int k = 0; k < kWgFwdBins / 64; ++k) sv[k] = src[64 * k + lane0];
    }
    // (while wave 0 builds the plan -- one global round trip -- the other waves build the twiddle tables)
    if (BANDK && band_on) { if (wave > 0) fft_build_twiddles_wg(twl, twh, tid - 64, (NW - 1) * 64); }
    else fft_build_twiddles_wg(twl, twh, tid, NW * 64);
    if (tid < kWgQueueInts) q[tid] = (tid == 0 && spec_pre) ? 1 : 0;
    if constexpr (STREAM) {
        for (int f = tid; f < p.F; f += NW * 64) coefT[f] = fin_coef(p.fin, f);
        if (tid < 8) sq[tid] = 0;
        for (int i = tid; i < RING * FPS; i += NW * 64) fr[i] = 0.0f;
    }
    FinCoef* lcoef = nullptr;                                             // (lds_sums) the filters' finalize coefficients, for the tail
    if (lds_sums) {
        const int n = (int)((long long)p.B * p.nblk / (int)gridDim.x / p.nblk) * p.F * p.TP;    // clips per workgroup x F x T'
        for (int i = tid; i < n; i += NW * 64) lsum[i] = 0.0f;
        lcoef = reinterpret_cast<FinCoef*>(lsum + (n + 7) / 8 * 8);
        for (int f = NW * 64 - 1 - tid; f < p.F; f += NW * 64) lcoef[f] = fin_coef(p.fin, f);   // (the last waves: wave 0 builds the plan)
    }
    __syncthreads();
#if LEAF_TRACE
    // phase stamps of workgroup 0, waves 0..7 (tools/trace_wg.py): tag << 56 | s_memtime
    int tr_n = 0;
#define WG_STAMP(tag)                                                                                           \
    do {                                                                                                        \
        if (LEAF_TRACE == 2 && (tag) != 1 && (tag) != 2 && (tag) < 8) break;   /* 2: task starts and finalize stamps only */ \
        if (blockIdx.x == 0 && wave < 16 && lane0 == 0 && tr_n < 64)                                            \
            p.trace[wave * 64 + tr_n] = ((unsigned long long)(tag) << 56) | (__builtin_amdgcn_s_memtime() & 0x00FFFFFFFFFFFFFFull); \
        ++tr_n;                                                                                                 \
    } while (0)
#else
#define WG_STAMP(tag) do { } while (0)
#endif

    constexpr int PADL = SK / 2 + SK % 2 - 1;
    constexpr int LS = fft_block_len(SK, SHOP, true);
    constexpr int DMIN = -((SK - 1 - PADL) / SHOP);
    constexpr int DMAX = (LS - 1 + PADL) / SHOP;
    constexpr int NFR = DMAX - DMIN + 1;
    constexpr int NROW = LS / 64;
    constexpr int NGRP = (NFR + 15) / 16;
    static_assert(LS % SHOP == 0 && LS % 64 == 0 && LS > 0 && NFR <= 32 && (SK & 1), "static odd-window geometry");
    static_assert(!STREAM || (LS + SK - PADL) / SHOP + 2 <= kWgOutChunk, "a block completes fewer frames than one output chunk");

    // Task ids: F + 1 slots per set (wg_task_decode); slot 0 of set i is fwd(i + 1), slots 1..F are the set's filters.
    // blocks are dealt CONTIGUOUSLY (workgroup w: blocks [first_gb, first_gb + nset)): a clip's blocks stay on one CU (the
    // overlap-save halo is re-read from this CU's cache) and a clip all of whose blocks this workgroup ran is finalized in
    // the tail below without another kernel
    const OwnedClips deal{p.B * p.nblk, (int)gridDim.x, p.nblk};
    if (spec_pre && wave == 1) {                                          // fwd(0): ring slot 0, generation 0
        const int gb0 = deal.start((int)blockIdx.x);
#pragma unroll
        for (int k = 0; k < kWgFwdBins / 64; ++k) ring[64 * k + lane0] = sv[k];
        if (lane0 == 0) { q[5] = gb0 / p.nblk; q[6] = gb0 % p.nblk; }
        wg_release();
        if (lane0 == 0) __hip_atomic_fetch_add(&q[1], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
#if LEAF_WG_STRIDED
    const int first_gb = (int)blockIdx.x;
    const int nset = (deal.nblocks - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x;
#else
    const int first_gb = deal.start((int)blockIdx.x);
    const int nset = deal.count((int)blockIdx.x);                         // blocks of this workgroup
#endif
    const int NT = band_on ? __builtin_amdgcn_readfirstlane(bl[0]) : p.F;   // filter tasks per block
    const int* tdesc = bl + kBandPlanHead;
    const int* bmem = tdesc + p.F + 4;
    (void)tdesc; (void)bmem;
    const WgTaskGrid grid = wg_task_grid(NT, nset);                        // NT + 1 slots per set
    const int ntasks = nset > 0 ? 1 + nset * (NT + 1) : 0;
    auto pull = [&]() {
        int v = 0;
        if (lane0 == 0) v = __hip_atomic_fetch_add(&q[0], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        return __builtin_amdgcn_readfirstlane(v);
    };
    // task -> (set, role): role 0 = forward transform of set `set`, role 1..F = filter role - 1 of set `set`
    auto decode = [&](int t, int& set, int& role) { wg_task_decode(grid, t, set, role); };
    // descriptor of a filter task: class (0: one filter on 2048 points, 1 / 2: band task) | index << 2 (filter / first member)
    auto desc_of = [&](int role) {
        if (role <= 0 || role > NT) return 0;
        return band_on ? __builtin_amdgcn_readfirstlane(tdesc[role - 1]) : (role - 1) << 2;
    };
#if LEAF_WG_PK
    static_assert(!HALF && LEAF_FFT32_DIT && LEAF_FFT_FUSE_TWIDDLE && LEAF_FFT_NOSWAP, "the packed front half is the 12-wave form of the default transform");
    v2f rqp[16];                                                          // (R_f[64 k + lane], R_f[64 (k + 16) + lane]): the operand pairs of the fused multiply
    auto load_real_spectrum = [&](int f, int lane) {
        const float* src = reinterpret_cast<const float*>(p.H) + (size_t)f * kFftN + lane;
        asm volatile("" ::: "memory");
#pragma unroll
        for (int k = 0; k < 16; ++k) { rqp[k].x = src[64 * k]; rqp[k].y = src[64 * (k + 16)]; }
        asm volatile("" ::: "memory");
    };
    v2f pkW[8], pk_two;                                                   // twiddle constants of the 32-point transform as register pairs
    pk_twiddle_pairs(pkW, pk_two);
#else
    float rq[32];                                                         // R_f[64 k + lane], natural row order
    auto load_real_spectrum = [&](int f, int lane) {
        const float* src = reinterpret_cast<const float*>(p.H) + (size_t)f * kFftN + lane;
        asm volatile("" ::: "memory");
#pragma unroll
        for (int k = 0; k < 32; ++k) rq[k] = src[64 * k];
        asm volatile("" ::: "memory");
    };
#endif
    // the 32 table values the task `role` starts with: a spectrum row, or the bins of a band task's windows (row 0 as a dummy
    // when there is no filter task)
    auto prefetch_task = [&](int role, int lane) {
        const int d = desc_of(role);
        if constexpr (BANDK && !LEAF_WG_PK) {
            if ((d & 3) == 1) { band_load_spectrum<16>(rq, reinterpret_cast<const float*>(p.H), bmem[(d >> 2) + lane / band_lpf(16)], lane); return; }
            if ((d & 3) == 2) { band_load_spectrum<32>(rq, reinterpret_cast<const float*>(p.H), bmem[(d >> 2) + lane / band_lpf(32)], lane); return; }
        }
        load_real_spectrum(d >> 2, lane);
    };

    // ---- STREAM: finalize the frames that set j's block completed (see the comment above the kernel); one wave, lane = filter
    auto stream_finalize = [&](int j) {
        // called by the wave that completed set j's last filter; blocks go out in order (EMA state, and set j - 1's sums)
        WG_STAMP(10);                                                     // finalize: waiting for the previous block's
        if (j > 0) wg_wait_ge(&q[11 + ((j - 1) & 1)], ((j - 1) >> 1) + 1);
        WG_STAMP(8);                                                      // finalize: start
        if (LEAF_STREAM_PRIO) __builtin_amdgcn_s_setprio(3);               // (see the switch: the launch's one dependent chain goes first)
        const int gb = first_gb + j;
        const int b = gb / p.nblk, c = gb - b * p.nblk;
        const int gbase = ((gb - first_gb) / p.nblk) * p.TP;              // ring numbering: clip ordinal x T' + m
        // frames final once block c is through: window end m hop - padL + K - 1 <= (c + 1) L - 1; the clip's last block
        // completes all that remain
        const int mf_prev = c == 0 ? -1 : min(p.TP - 1, (c * LS - SK + PADL) / SHOP);
        const int mf = c == p.nblk - 1 ? p.TP - 1 : min(p.TP - 1, ((c + 1) * LS - SK + PADL) / SHOP);
        const int mode = p.fin.mode;
        const bool scaled = p.fin.clip_scale2 != nullptr;
        const float s2 = scaled ? p.fin.clip_scale2[b] : 1.0f;
#ifndef LEAF_STREAM_ABLATE
#define LEAF_STREAM_ABLATE 0       // measurement only (results wrong): 1 = the finalize loop is skipped, 2 = only its stores are
#endif
        // Four frames at a time: (A) the slots, the pooled values and the EMA recurrence -- the only sequential part, three
        // operations per frame -- into registers; (B) four INDEPENDENT point functions, straight-line, so that the scheduler
        // interleaves their dependent chains (log2 / exp2 / rcp at 16+ cycles each): finalized one frame after the other, a
        // lone wave on a SIMD shared with two transform waves took ~25 cycles per instruction, 4 task-times per block.
        constexpr int G = 4;
        for (int f0 = 0; f0 < p.F && LEAF_STREAM_ABLATE != 1; f0 += 64) {
            const int f = f0 + lane0;
            const bool on = f < p.F;
            const FinCoef cf = coefT[on ? f : 0];
            const bool fast = __all(!on || cf.dl > 0.0f);                 // every filter on the cancellation-free PCEN form
            float M = (c == 0 || !on) ? 0.0f : ema_st[f];
            const size_t orow = ((size_t)b * p.F + (on ? f : 0)) * p.TP;
            for (int base = mf_prev + 1; base <= mf; base += G) {
                const int cnt = min(G, mf - base + 1);
                float sa[G], sb[G], v[G], Mv[G];
                int ns[G];
#pragma unroll
                for (int k = 0; k < G; ++k) {                            // (A) loads first
                    const int m = min(base + k, mf);
                    // the frame's accumulator: a (+ b when the window meets two blocks: added in LDS, a + b either way round -- the
                    // rounding of `part`'s slot 0 + slot 1); read, and zero for the frame that wraps onto it
                    float* e = fr + (size_t)((gbase + m) & (RING - 1)) * FPS + (on ? f : 0);
                    ns[k] = 1;
                    sa[k] = e[0];
                    sb[k] = 0.0f;
                    if (on && k < cnt) e[0] = 0.0f;
                }
#pragma unroll
                for (int k = 0; k < G; ++k) {                            // ... then the recurrence, in frame order
                    const int m = base + k;
                    float x = fin_pooled(sa[k], sb[k], 0.0f, ns[k], scaled, s2, cf.bias);
                    if (k < cnt && on && p.fin.raw_out) p.fin.raw_out[orow + m] = x;
                    if (!(mode & 8)) x = pooled_floor(x);
                    v[k] = x;
                    if ((mode & 1) && k < cnt) {
                        if (m == 0) M = x;                                // state starts at p_0 (postprocessing.py:15)
                        M = fin_ema_step(cf, x, M);
                    }
                    Mv[k] = M;
                }
                if (fast && (mode & 1)) {                                 // (B) branch-free: the chains of the eight frames interleave
                    float o[G];
#pragma unroll
                    for (int k = 0; k < G; ++k) o[k] = fin_point_pcen_pos(cf, p.fin.floor_, v[k], Mv[k]);
#pragma unroll
                    for (int k = 0; k < G; ++k)
                        if (k < cnt && on && (LEAF_STREAM_ABLATE != 2 || o[k] == 12345.678f))
                            ost[f * kWgOutRow + ((base + k) & (2 * kWgOutChunk - 1))] = o[k];
                } else {
                    // PCEN off, or a filter with delta <= 0 (the reference's literal powf form): one compact out-of-line
                    // point function, frame after frame -- rare, and the code of this kernel has to stay inside the
                    // instruction cache next to the transforms (87 KB with this path unrolled: the whole kernel ran 15 % slower)
#pragma unroll
                    for (int k = 0; k < G; ++k) {
                        const float ov = fin_point_outofline(cf, mode, p.fin.floor_, v[k], Mv[k]);
                        if (k < cnt && on) ost[f * kWgOutRow + ((base + k) & (2 * kWgOutChunk - 1))] = ov;
                    }
                }
            }
            if (on) ema_st[f] = M;
        }
        // ---- flush: every 32-frame chunk of the clip's rows whose last frame is now final (the clip's last block: also the
        // partial chunk at its end).  Chunks q with (q + 1) 32 - 1 <= mf_prev went out with an earlier block.
        if (LEAF_STREAM_ABLATE != 1) {
            const int q0 = (mf_prev + 1) / kWgOutChunk;
            const int q1 = c == p.nblk - 1 ? (p.TP - 1) / kWgOutChunk : (mf + 1) / kWgOutChunk - 1;   // last chunk that is complete
            for (int qc = q0; qc <= q1; ++qc) {
                const int m = qc * kWgOutChunk + (lane0 & (kWgOutChunk - 1));
                for (int r0 = 0; r0 < p.F; r0 += 64 / kWgOutChunk) {
                    const int row = r0 + lane0 / kWgOutChunk;
                    if (row < p.F && m < p.TP)
                        fin_store(p.fin, ((size_t)b * p.F + row) * p.TP + m, ost[row * kWgOutRow + (m & (2 * kWgOutChunk - 1))]);
                }
            }
        }
        if (LEAF_STREAM_PRIO) __builtin_amdgcn_s_setprio(0);
        WG_STAMP(9);                                                      // finalize: done
        wg_release();                // ring entries read, EMA state written: block j is out
        if (lane0 == 0) __hip_atomic_fetch_add(&q[11 + (j & 1)], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    };
    (void)stream_finalize;

    // Invariant at the loop head: (set, role) is the decoded current task, and when it is an inverse task its filter's
    // spectrum row has already been requested into rq (by the previous task, under its pooling).
    int seen_set = -1, seen_b = 0, seen_c = 0, seen_base = 0, seen_clip = 0;   // block coordinates of the set this wave last worked on
    int t = pull(), set = 0, role = 0;
    if (t < ntasks) decode(t, set, role);
    prefetch_task(role, lane0);
    while (t < ntasks) {
        int lane = lane0;
        asm volatile("" : "+v"(lane));
        const int slot = set & 1, gen = set >> 1;
        float2* A = ring + slot * kWgRingFwdFloat2;
        WG_STAMP(role == 0 ? 1 : 2);                                      // task taken: 1 forward transform, 2 filter
        if (role == 0) {
            // ---- forward transform of block gb into ring slot `slot` (skipped past the last set)
            if (set < nset) {
                const int gb = first_gb + set * (LEAF_WG_STRIDED ? (int)gridDim.x : 1);
                const int b = gb / p.nblk, c = gb - b * p.nblk;               // the only division per block
                const int n_c = c * LS;
                float are[32], aim[32];
                const float* xb = static_cast<const float*>(p.x) + (size_t)b * p.T;
                const unsigned short* xh = static_cast<const unsigned short*>(p.x) + (size_t)b * p.T;
                if (p.io_bf16) {
#pragma unroll
                    for (int r = 0; r < 32; ++r) {
                        const int i = 64 * r + lane;                      // block rotated left by padL samples
                        const int n = n_c - PADL + ((i + PADL) & (kFftN - 1));
                        const unsigned v = xh[min(max(n, 0), p.T - 1)];
                        are[r] = (n >= 0 && n < p.T) ? __uint_as_float(v << 16) : 0.0f;
                        aim[r] = 0.0f;
                    }
                } else {
#pragma unroll
                    for (int r = 0; r < 32; ++r) {
                        const int i = 64 * r + lane;
                        const int n = n_c - PADL + ((i + PADL) & (kFftN - 1));
                        are[r] = (n >= 0 && n < p.T) ? xb[n] : 0.0f;
                        aim[r] = 0.0f;
                    }
                }
                fft2048w<HALF>(are, aim, scr, scr_lds, twl, twh, lane);        // register i <-> bin 64 brev5(i) + lane
                wg_wait_ge(&q[3 + slot], gen * NT);                       // the slot's previous readers are done
                if constexpr (STREAM) {                                   // ... and the frames ours wrap onto in the ring are out
                    const int lag = wg_stream_lag(SK, SHOP, RING);        // (block set - lag: same parity, (lag / 2) generations back)
                    if (set >= lag) wg_wait_ge(&q[11 + slot], gen - lag / 2 + 1);
                }
#pragma unroll
                for (int i = 0; i < 32; ++i) {
                    const int k = brev5(i);
                    if (k < kWgFwdBins / 64) A[64 * k + lane] = make_float2(are[i], aim[i]);     // bins 0..1151: 1025.. for band windows that cross Nyquist
                }
                if (lane == 0) { q[5 + 2 * slot] = b; q[6 + 2 * slot] = c; }      // the block's coordinates, for its readers
                wg_release();
                if (lane == 0) __hip_atomic_fetch_add(&q[1 + slot], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
            // rq is redefined UNCONDITIONALLY here (row 0 when the next task is not an inverse one), so that the previous
            // row is dead throughout this branch -- carried through the forward transform it would be spilled every task
            t = pull();
            if (t < ntasks) decode(t, set, role);
            else role = 0;
            prefetch_task(role, lane);
            continue;
        }
        // ---- filter task `role` of the block in ring slot `slot`
        const int tdsc = desc_of(role);
        const int f = tdsc >> 2;
        if (set != seen_set) {                                            // this wave's first filter of the block: once the
            wg_wait_ge(&q[1 + slot], gen + 1);                            // spectrum is in the ring it stays until every filter is done
            seen_b = __builtin_amdgcn_readfirstlane(wg_ld(&q[5 + 2 * slot]));
            seen_c = __builtin_amdgcn_readfirstlane(wg_ld(&q[6 + 2 * slot]));
            if constexpr (STREAM) seen_base = ((set - seen_c) / p.nblk) * p.TP;   // clip ordinal in this workgroup x T' (aligned dealing)
            seen_clip = lds_sums ? (set - seen_c) / p.nblk : 0;           // clip ordinal in this workgroup (whole clips per workgroup)
            seen_set = set;
        }
        WG_STAMP(3);                                                      // spectrum available
        const int b = seen_b, c = seen_c;
        const int n_c = c * LS;
        const int Lv = min(LS, p.T - n_c);
        int mlo = n_c + PADL - SK + 1;                                    // first frame whose window reaches the block
        mlo = mlo <= 0 ? 0 : (mlo + SHOP - 1) / SHOP;
        const int mhi = min(p.TP - 1, (n_c + Lv - 1 + PADL) / SHOP);
        if constexpr (BANDK) {
            if (tdsc & 3) {
                // ---- band task: eight (four) narrow-band filters on 256- (512-) point transforms (leaf_band.hpp)
                int tn_b = 0, nset_b = 0, nrole_b = 0;
                auto mid = [&]() {
                    tn_b = pull();
                    if (tn_b < ntasks) decode(tn_b, nset_b, nrole_b);
                    prefetch_task(nrole_b, lane);
                    return std::integral_constant<int, 32>{};             // 32 loads issued
                };
                auto stamp = [&](int tag) { (void)tag; WG_STAMP(tag); };
                // the block's share of frame m of filter `fid`: where the 2048-point task puts it
                auto bout = [&](int fid, int m, float v) {
                    const int first_block = max(0, m * SHOP - PADL) / LS;
                    if constexpr (STREAM)
                        __hip_atomic_fetch_add(&fr[(size_t)((seen_base + m) & (RING - 1)) * FPS + fid], v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    else if (lds_sums)
                        __hip_atomic_fetch_add(&lsum[((size_t)seen_clip * p.F + fid) * p.TP + m], v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    else
                        p.part[(((size_t)b * p.F + fid) * p.nslot + (c - first_block)) * p.TP + m] = v;
                };
                if ((tdsc & 3) == 1)
                    band_task<16, SK, SHOP>(p, rq, A, bmem + (tdsc >> 2), bl + 4, twl, scr, scr_lds, &q[3 + slot], c, mlo, mhi, lane, mid, bout, stamp);
                else
                    band_task<32, SK, SHOP>(p, rq, A, bmem + (tdsc >> 2), bl + 4, twl, scr, scr_lds, &q[3 + slot], c, mlo, mhi, lane, mid, bout, stamp);
                if constexpr (STREAM) {                                   // as below: the block's last task sends its frames out
                    int done = 0;
                    if (lane == 0) done = __hip_atomic_fetch_add(&sq[set & 7], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    done = __builtin_amdgcn_readfirstlane(done);
                    if (done + 1 == ((set >> 3) + 1) * NT) stream_finalize(set);
                }
                WG_STAMP(7);
                t = tn_b;
                set = nset_b;
                role = nrole_b;
                continue;
            }
        }
        // Z = conj(A' R_f): rows 0..15 straight from the ring, rows 16..31 mirrored (A'[N - e] = conj(A'[e]))
        // (8-row chunks, fenced: all 32 ring reads in flight at once would need 64 registers next to rq and Z)
        float zre[32], zim[32];
#if LEAF_WG_PK
        v2f zc[32];                                                       // Z as complex pairs (zre / zim are filled by the transform's scalar back half)
#endif
        {
            // two streams of 16 rows: ascending from A[lane], and the mirror A[2048 - 64 k - lane], k = 16..31, read as
            // rows 15..0 of the base A[1088 - lane] (= k = 31 first); lds_stream32 walks 2 x 16 rows
            const unsigned a_lo = lds_addr(A + lane), a_hi = lds_addr(A + (kFftN - 64 * 31) - lane);
            v2f lo[16], hi[16];
            auto rd = [&](auto kk) {
                constexpr int k = decltype(kk)::value;
                if constexpr (k < 16) lds_rd8<512 * k>(lo[k], a_lo);
                else lds_rd8<512 * (31 - k)>(hi[k - 16], a_hi);
            };
            (void)rd;
            // chunk 0: rows 0..7, chunk 1: rows 8..15, chunk 2: rows 16..23, chunk 3: rows 24..31
#define LEAF_RD8(B) rd(std::integral_constant<int, B + 0>{}); rd(std::integral_constant<int, B + 1>{}); \
                    rd(std::integral_constant<int, B + 2>{}); rd(std::integral_constant<int, B + 3>{}); \
                    rd(std::integral_constant<int, B + 4>{}); rd(std::integral_constant<int, B + 5>{}); \
                    rd(std::integral_constant<int, B + 6>{}); rd(std::integral_constant<int, B + 7>{});
            v2f(&lo0)[8] = *reinterpret_cast<v2f(*)[8]>(&lo[0]);
            v2f(&lo1)[8] = *reinterpret_cast<v2f(*)[8]>(&lo[8]);
            v2f(&hi0)[8] = *reinterpret_cast<v2f(*)[8]>(&hi[0]);
            v2f(&hi1)[8] = *reinterpret_cast<v2f(*)[8]>(&hi[8]);
#if LEAF_WG_PK
            // the fused multiply on complex pairs: three packed instructions per pair of rows instead of six scalar ones
            auto pair = [&](int k) {
                v2f t;
                asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel:[0,0] op_sel_hi:[1,0] neg_hi:[1,0]" : "=v"(t) : "v"(lo[k]), "v"(rqp[k]));   // (a.x ra, -a.y ra)
                asm volatile("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,1,0] op_sel_hi:[1,1,1]" : "=v"(zc[k]) : "v"(hi[k]), "v"(rqp[k]), "v"(t));
                asm volatile("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,1,0] op_sel_hi:[1,1,1] neg_lo:[1,0,0] neg_hi:[1,0,0]"
                             : "=v"(zc[k + 16]) : "v"(hi[k]), "v"(rqp[k]), "v"(t));
            };
            LEAF_RD8(0) LEAF_RD8(16) LEAF_RD8(8)
            lds_wait8<8>(lo0);
            lds_wait8<8>(hi0);
#pragma unroll
            for (int k = 0; k < 8; ++k) pair(k);
            LEAF_RD8(24)
            lds_wait8<0>(lo1);
            lds_wait8<0>(hi1);
#pragma unroll
            for (int k = 8; k < 16; ++k) pair(k);
#elif LEAF_FFT32_DIT && LEAF_FFT_FUSE_TWIDDLE
            // the spectral multiply fused with the first decimation-in-time stage of the transform (pairs of rows (k, k + 16),
            // unit twiddles): with za = conj(A'[k]) R[k] and zb = the mirrored row's product,
            //     out[k] = za + zb,  out[k + 16] = za - zb   as one product and two FMAs per component -- 6 instructions per pair
            // instead of 4 products + 4 additions.  Mirrored rows are read as rows 15..0 of a_hi: row k + 16 is hi[k].
            auto pair = [&](int k) {
                const float ra = rq[k], rb = rq[k + 16];
                const float tr_ = lo[k].x * ra, ti_ = -(lo[k].y * ra);
                zre[k] = fmaf(hi[k].x, rb, tr_);
                zim[k] = fmaf(hi[k].y, rb, ti_);
                zre[k + 16] = fmaf(-hi[k].x, rb, tr_);
                zim[k + 16] = fmaf(-hi[k].y, rb, ti_);
            };
            LEAF_RD8(0) LEAF_RD8(16) LEAF_RD8(8)
            lds_wait8<8>(lo0);
            lds_wait8<8>(hi0);
#pragma unroll
            for (int k = 0; k < 8; ++k) pair(k);
            LEAF_RD8(24)
            lds_wait8<0>(lo1);
            lds_wait8<0>(hi1);
#pragma unroll
            for (int k = 8; k < 16; ++k) pair(k);
#else
            LEAF_RD8(0) LEAF_RD8(8)
            lds_wait8<8>(lo0);
#pragma unroll
            for (int k = 0; k < 8; ++k) { zre[k] = lo[k].x * rq[k]; zim[k] = -(lo[k].y * rq[k]); }
            LEAF_RD8(16)
            lds_wait8<8>(lo1);
#pragma unroll
            for (int k = 8; k < 16; ++k) { zre[k] = lo[k].x * rq[k]; zim[k] = -(lo[k].y * rq[k]); }
            LEAF_RD8(24)
            lds_wait8<8>(hi0);
#pragma unroll
            for (int k = 16; k < 24; ++k) { zre[k] = hi[k - 16].x * rq[k]; zim[k] = hi[k - 16].y * rq[k]; }
            lds_wait8<0>(hi1);
#pragma unroll
            for (int k = 24; k < 32; ++k) { zre[k] = hi[k - 16].x * rq[k]; zim[k] = hi[k - 16].y * rq[k]; }
#endif
#undef LEAF_RD8
        }
#if LEAF_WG_PK
        asm volatile("s_waitcnt lgkmcnt(0)" ::"v"(zc[31]) : "memory");
#else
        asm volatile("s_waitcnt lgkmcnt(0)" ::"v"(zre[31]), "v"(zim[31]) : "memory");
#endif
        wg_release();
        if (lane == 0) __hip_atomic_fetch_add(&q[3 + slot], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        // pooling row of this filter -> wave-private LDS (16 bytes per lane per instruction), lands under the transform
        if constexpr (!LEAF_WG_REGW) {
            const float* gsrc = p.Gz + (size_t)f * p.GZ;
#pragma unroll
            for (int i0 = 0; i0 < GU; i0 += 256)
                if (i0 + 256 <= GU || i0 + 4 * lane < GU)
                    __builtin_amdgcn_global_load_lds(gsrc + i0 + 4 * lane, (__attribute__((address_space(3))) void*)(sG + i0), 16, 0, 0);
            asm volatile("" ::: "memory");
        }
        WG_STAMP(4);                                                      // spectral multiply done
#if LEAF_WG_PK
        fft2048w_pkfront(zc, zre, zim, scr, scr_lds, twl, twh, lane, pkW, pk_two);
#else
        fft2048w<HALF, LEAF_FFT32_DIT && LEAF_FFT_FUSE_TWIDDLE>(zre, zim, scr, scr_lds, twl, twh, lane);   // register i <-> samples 64 brev5(i) + lane
#endif
        WG_STAMP(5);                                                      // inverse transform done
#if LEAF_WG_REGW
        // the filter's pooling weights, NJ vectors (see wg_pool_nj): requested now, consumed after the energies
        constexpr int PG = wg_pool_step(SHOP), PJ0 = wg_pool_jmin(SK, SHOP), NJ = wg_pool_nj(SK, SHOP);
        float pw[NJ];
        {
            const float* gsrc = p.Gz + (size_t)f * p.GZ + (kGPad + PJ0) + lane;
            asm volatile("" ::: "memory");
#pragma unroll
            for (int k = 0; k < NJ; ++k) pw[k] = gsrc[PG * k];
            asm volatile("" ::: "memory");
        }
#endif
        float er[NROW];
#pragma unroll
        for (int i = 0; i < 32; ++i) {
            const int r = brev5(i);
            if (r < NROW) er[r] = zre[i] * zre[i] + zim[i] * zim[i];
        }
        if (Lv < LS) {                                                    // a clip's last block: outputs past the clip's end
#pragma unroll
            for (int r = 0; r < NROW; ++r) er[r] = 64 * r + lane < Lv ? er[r] : 0.0f;
        }
        // next task: reserved now so that its filter's spectrum row streams in under the pooling
        const int tn = pull();
        int nset_i = 0, nrole = 0;
        if (tn < ntasks) decode(tn, nset_i, nrole);
        asm volatile("" ::"v"(er[0]), "v"(er[NROW - 1]));
        prefetch_task(nrole, lane);                                      // (row 0 as a dummy when there is no next filter)
        asm volatile("s_waitcnt vmcnt(32)" ::: "memory");                 // the pooling weights (issued before the 32 loads) have landed
        WG_STAMP(6);                                                      // energies, next task reserved, pooling row landed
        float acc[NGRP][16];
#pragma unroll
        for (int g = 0; g < NGRP; ++g)
#pragma unroll
            for (int fi = 0; fi < 16; ++fi) acc[g][fi] = 0.0f;
#pragma unroll
        for (int r = 0; r < NROW; ++r) {
#pragma unroll
            for (int fi = 0; fi < NFR; ++fi) {
                const int is = (DMIN + fi) * SHOP - PADL;
                if (is <= 64 * r + 63 && is + SK > 64 * r) {
#if LEAF_WG_REGW
                    static_assert((PADL - PJ0) % PG == 0, "window offsets are congruent to padL modulo gcd(64, hop)");
                    acc[fi / 16][fi % 16] = fmaf(er[r], pw[(64 * r - is - PJ0) / PG], acc[fi / 16][fi % 16]);
#else
                    acc[fi / 16][fi % 16] = fmaf(er[r], sG[kGPad + 64 * r - is + lane], acc[fi / 16][fi % 16]);
#endif
                }
            }
        }
        asm volatile("" : "+v"(acc[0][0]));
#pragma unroll
        for (int g = 0; g < NGRP; ++g) {
            const float v = frame_butterfly16(acc[g], lane);
            const int fi = 16 * g + ((lane >> 5) & 1) * 8 + ((lane >> 4) & 1) * 4 + ((lane >> 3) & 1) * 2 + ((lane >> 2) & 1);
            const int m = n_c / SHOP + DMIN + fi;
            if ((lane & 3) == 0 && fi < NFR && m >= mlo && m <= mhi) {
                const int first_block = max(0, m * SHOP - PADL) / LS;
                if constexpr (STREAM)
                    __hip_atomic_fetch_add(&fr[(size_t)((seen_base + m) & (RING - 1)) * FPS + f], v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                else if (lds_sums)
                    __hip_atomic_fetch_add(&lsum[((size_t)seen_clip * p.F + f) * p.TP + m], v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                else
                    p.part[(((size_t)b * p.F + f) * p.nslot + (c - first_block)) * p.TP + m] = v;
            }
        }
        if constexpr (STREAM) {
            // This filter's frame sums are in the ring (LDS operations of a wave execute in order, so the count below lands
            // after them); the count's return value says whether this was the block's last filter, and if so the block's
            // frames go out NOW, before this wave can block on anything else.  (Looking at the count a task later saves the LDS
            // round trip per task, but a pending finalize on a wave that then waits for a spectrum whose forward task waits
            // for that very finalize is a deadlock -- it happened at F = 3, where a wave's next task is several blocks ahead.)
            int done = 0;
            if (lane == 0) done = __hip_atomic_fetch_add(&sq[set & 7], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            done = __builtin_amdgcn_readfirstlane(done);
            if (done + 1 == ((set >> 3) + 1) * NT) stream_finalize(set);
        }
        WG_STAMP(7);                                                      // pooling, reduction and stores issued
        // the pooling's LDS reads of sG must be complete before the next task's row DMA overwrites the buffer
        if constexpr (!LEAF_WG_REGW) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        t = tn;
        set = nset_i;
        role = nrole;
    }
    // ---- clip-resident finalize: every partial sum of a clip this workgroup owns outright was written by its own waves.
    // Release (the stores have reached L2) - barrier - then one wave per pair of (clip, filter) rows: bias, floor, EMA scan,
    // PCEN (fft_finalize_rows; the partial sums are read past the vector cache).  Clips that straddle two workgroups are
    // left to fft_finalize_kernel.
    if (!STREAM && LEAF_WG_TAIL && !LEAF_WG_STRIDED && p.fin_fused) {
        const int b_lo = (first_gb + p.nblk - 1) / p.nblk, b_hi = (first_gb + nset) / p.nblk;
        // <= 64 filters' rows per tile
        constexpr int TR = (HALF || NW < 10) ? 32 : 64;                    // (fewer waves -- A/B builds -- hold a smaller tile)
        static_assert((size_t)NW * SCRF >= (size_t)fin_tile_floats<TR, 64>() && (size_t)NW * SCRF >= (size_t)fin_tile_floats_single<TR, 128>(),
                      "the transposition scratch of all waves holds a finalize tile");
        FinParams fin = p.fin;
        if (lds_sums) {
            fin.lds_sums = lsum;
            fin.lds_row0 = b_lo * p.F;
            fin.lds_coef = lcoef;
        }
        wg_tail_finalize<TR>(fin, b_lo, b_hi, reinterpret_cast<float*>(q + kWgQueueInts), tid, NW * 64);   // every task is done: the scratch is free
    }
}

}  // namespace
