// leaf_kernels.hip -- MI355X (gfx950 / CDNA4) kernels for the LEAF frontend forward path, and the
// C ABI declared in include/leaf_hip.h.  Written for gfx950 only: 64-lane wavefronts, fp32 MFMA
// (v_mfma_f32_16x16x4_f32), 160 KiB LDS per CU.  No torch types anywhere in this file.
//
// Reference arithmetic being replaced (file:line under the reference repository):
//   leaf_pytorch/frontend.py:78-89        Leaf.forward
//   leaf_pytorch/convolution.py:15-22     GaborConstraint            -> constrain() below
//   leaf_pytorch/impulse_responses.py:5-16,66-71  Gabor taps         -> gabor_tap()
//   leaf_pytorch/convolution.py:71-99     GaborConv1d.forward        -> fused kernel / conv_staged
//   leaf_pytorch/frontend.py:15-19        SquaredModulus             -> fused kernel / sqmod
//   leaf_pytorch/impulse_responses.py:74-80 + pooling.py:31-42       -> fused kernel / pool_staged
//   leaf_pytorch/postprocessing.py:13-28,62-69  EMA + PCEN           -> finalize kernel
//
// Design (the derivations and the measured ladders of rounds 1-3 are in NOTES.md).  Three interchangeable formulations of the same arithmetic, chosen
// per call (LEAF_ALGO_AUTO):
//   * leaf_fft*.hpp -- overlap-save FFT (default for windows from 224 taps): a wave-level 2048-point FFT (32 x 64
//     four-step: register butterflies, LDS transposition, a swap-free half-wave step), the block's spectrum shared by the
//     filters; per filter a spectral multiply (real spectrum; an even window's unpaired tap in the time domain), one
//     inverse transform, |.|^2 and the Gaussian pooling to per-frame partial sums; a row kernel sums the partials, floors
//     at 1e-5 and runs the EMA/PCEN scan.  With a backward epilogue the same structure is the backward (tap gradient = two
//     spectral dot products per block and filter; dL/dx = one more transform per block).
//       leaf_fft.hpp              one wave per block (small batches; round 1's kernel)
//       leaf_fft_wg.hpp / _bwd    one workgroup per block, spectrum ring in LDS, task queue: static LEAF geometries
//       leaf_fft_wg4k.hpp         4096-sample blocks for K = 801 / hop 320 (two half transforms per filter)
//       leaf_fft_wgg.hpp / _bwd   the same with window, hop and block length at run time (64..1216 taps, odd or even)
//       leaf_fft_wgg4k.hpp / _bwd 4096-sample blocks with run-time geometry (odd windows 833..2049)
//   * leaf_fused.hpp -- direct form on the fp32 MFMA (short windows, and the backward for even / short windows):
//     the Gabor taps are Hermitian in t (Re even, Im odd), so with s_k[n] = x[n+k] + x[n-k] and
//     d_k[n] = x[n+k] - x[n-k] the complex filterbank is two real GEMMs with HALF the K extent,
//         Re y[n,f] = sum_k s_k[n] * hr_f[k],   Im y[n,f] = sum_k d_k[n] * hi_f[k],  k = 0..K/2,
//     exact fp32 fmaf chains at the fp32 vector rate with operands delivered from LDS; one wave owns one hop-block,
//     squares its accumulator tiles, applies the pooling weights and reduces to per-frame partial sums.
//   * leaf_staged.hpp -- one kernel per reference module, every intermediate materialised: the sub-modules' own
//     forwards, the on-device cross-check of the fused kernels, and the fallback for geometries nothing else covers.
//   In the fused paths the 80x-inflated (B,2F,T) tensor of the reference never exists.
#include <atomic>
#include <random>
#include <cstdlib>
#include "leaf_common.hpp"
#include "leaf_staged.hpp"
#include "leaf_fused.hpp"
#include "leaf_backward.hpp"
#include "leaf_stage_backward.hpp"
#include "leaf_fft.hpp"
#include "leaf_fft_wg.hpp"
#include "leaf_fft_small.hpp"
#include "leaf_fft_wg_bwd.hpp"
#include "leaf_fft_wg4k.hpp"
#include "leaf_fft_wgg.hpp"
#include "leaf_fft_wgg_bwd.hpp"
#include "leaf_fft_wgg4k.hpp"
#include "leaf_fft_wgg4k_bwd.hpp"
#include "leaf_inst.hpp"
namespace {

// The big kernel templates are instantiated in the inst_*.hip translation units (leaf_inst.hpp); here they are opaque handles.
using FftKernel = void (*)(const FftParams);
inline FftKernel as_fft_kernel(const void* h) { return reinterpret_cast<FftKernel>(const_cast<void*>(h)); }
// Tools-only environment switches (A/B measurements of tools/*.py): compiled in only with -DLEAF_TOOLS=1
// (LEAF_HIPCC_EXTRA="-DLEAF_TOOLS=1"); the product library reads no environment variable except LEAF_NO_4K.
inline const char* tools_env(const char* name) { return LEAF_TOOLS ? getenv(name) : nullptr; }

// ---------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------

inline int ceil_div(int a, int b) { return (a + b - 1) / b; }
inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

// LEAF_ALGO_RESERVE_CUS(k): CUs the current call leaves free (thread-local and scoped to one ABI call by ReserveCus: the
// library stays re-entrant and keeps nothing between calls)
thread_local int tl_reserved_cus = 0;
thread_local bool tl_stream_finalize = false;                // LEAF_ALGO_STREAM_FINALIZE of the current call
thread_local bool tl_band_off = false;                       // LEAF_ALGO_FULL_TRANSFORMS of the current call
thread_local bool tl_band_strict = false;                    // LEAF_ALGO_STRICT_BAND_CLASSES of the current call
struct ReserveCus {
    int prev;
    bool prev_stream, prev_band_off, prev_band_strict;
    explicit ReserveCus(int algo) : prev(tl_reserved_cus), prev_stream(tl_stream_finalize), prev_band_off(tl_band_off), prev_band_strict(tl_band_strict) {
        if ((algo >> 16) & 0xff) tl_reserved_cus = (algo >> 16) & 0xff;   // nested calls pass the masked selector: they inherit
        if (algo & LEAF_ALGO_STREAM_FINALIZE) tl_stream_finalize = true;
        if (algo & LEAF_ALGO_FULL_TRANSFORMS) tl_band_off = true;
        if (algo & LEAF_ALGO_STRICT_BAND_CLASSES) tl_band_strict = true;
    }
    ~ReserveCus() { tl_reserved_cus = prev; tl_stream_finalize = prev_stream; tl_band_off = prev_band_off; tl_band_strict = prev_band_strict; }
};
int device_cus();
// CUs this call may fill: the device's count minus the call's reservation (at least one)
int num_cus() { return std::max(1, device_cus() - tl_reserved_cus); }
// CU count of the CURRENT device (cached per device ordinal; plans are sized for the device the call runs on)
int device_cus() {
    constexpr int kMaxDev = 64;
    static std::atomic<int> cached[kMaxDev];
    int dev = 0, n = 0;
    if (hipGetDevice(&dev) != hipSuccess) return 256;
    if (dev >= 0 && dev < kMaxDev) {
        n = cached[dev].load(std::memory_order_relaxed);
        if (n > 0) return n;
    }
    if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
    if (dev >= 0 && dev < kMaxDev) cached[dev].store(n, std::memory_order_relaxed);
    return n;
}

struct FusedPlan {
    bool ok;
    int FP, ntiles, KS, R, Hf, xshift, NBH, NU, HP, XS, q_lo, q_hi, nq, noff, noff_t, padL, TP;
    int rt_main, groups_main, rt_rem;
    int GJ;
    size_t w_floats, g_floats, part_floats, meta_ints;
};

// +4 tap rows and +16 window floats per wave: the k-loop prefetches one step past its last k-step
inline size_t fused_lds_bytes(int R, int rt, int XS) {
    return ((size_t)(R + 4) * 32 * rt + (size_t)kWavesPerWG * (XS + 16)) * 4;
}

FusedPlan make_plan(int B, int T, int F, int K, int hop) {
    FusedPlan pl{};
    pl.padL = K / 2 + K % 2 - 1;
    pl.TP = (T + (K - 1) - K) / hop + 1;
    pl.FP = 16 * ceil_div(F, 16);
    pl.ntiles = pl.FP / 16;
    pl.KS = ceil_div(K / 2 + 1, 4);
    pl.R = 4 * pl.KS;
    pl.Hf = (K - 1) / 2;
    pl.xshift = K / 2 - pl.padL;
    pl.NBH = ceil_div(hop, 16);
    pl.NU = ceil_div(pl.NBH, kUB);
    pl.HP = 4 * pl.KS;
    pl.XS = 16 * kUB * pl.NU + 2 * pl.HP;
    pl.q_lo = pl.padL / hop;
    pl.q_hi = (T - 1 + pl.padL) / hop;
    pl.nq = pl.q_hi - pl.q_lo + 1;
    pl.noff = (K - 1) / hop + 1;
    pl.noff_t = pl.noff <= 1 ? 1 : (pl.noff <= 3 ? 3 : (pl.noff <= 6 ? 6 : 0));
    pl.w_floats = (size_t)pl.R * 2 * pl.FP;
    pl.GJ = ((std::max(pl.noff_t, 1) - 1) * hop + 16 * kUB * pl.NU + 3) / 4 * 4;
    pl.g_floats = (size_t)pl.FP * pl.GJ;
    pl.part_floats = (size_t)B * pl.TP * pl.noff * pl.FP;
    pl.meta_ints = (size_t)2 * pl.FP + pl.ntiles;          // perm[FP], col_of[FP], tile_ks[ntiles]
    pl.rt_main = 0;
    // Widest register tile whose taps fit LDS -- unless the batch is so small that (tasks x filter groups) would not
    // even give every wave slot of the chip one task: then narrower tiles (more filter groups, shorter per-wave
    // dependency chains) cut the latency of a small forward.
    const long long wave_slots = (long long)num_cus() * kWavesPerWG;
    // NOFF = 6 instances (4..6 overlapping frames per hop-block): the widest register tiles spill (720 B/lane at RT = 3,
    // 188 B at RT = 2, -Rpass-analysis=kernel-resource-usage), so the tile is capped where it stays in registers;
    // LEAF_FUSED_RT_CAP (environment, tools only) overrides the cap for measurements.
    static const int rt_cap_env = [] { const char* e = tools_env("LEAF_FUSED_RT_CAP"); return e ? atoi(e) : 0; }();
    const int rt_cap = rt_cap_env > 0 ? rt_cap_env : (pl.noff_t == 6 ? 1 : 3);
    for (int rt = 3; rt >= 1; --rt) {
        if (rt > rt_cap && rt > 1) continue;
        if (rt > pl.ntiles || fused_lds_bytes(pl.R, rt, pl.XS) > (size_t)kMaxLds) continue;
        if (pl.rt_main == 0) pl.rt_main = rt;
        if ((long long)B * pl.nq * ceil_div(pl.ntiles, rt) >= wave_slots) break;
        pl.rt_main = rt;
    }
    pl.ok = pl.rt_main > 0 && pl.noff_t > 0 && pl.FP <= kMaxFP && (long long)B * pl.nq < (1ll << 30) &&
            (double)pl.part_floats < 2.0e9;
    if (pl.ok) {
        pl.groups_main = pl.ntiles / pl.rt_main;
        pl.rt_rem = pl.ntiles % pl.rt_main;
    }
    return pl;
}

hipError_t launch_fused(const FusedParams& prm, int rt, int noff_t, int groups, size_t lds, int grid_x, hipStream_t st) {
    const void* h = leaf_inst_fused(rt, noff_t, (prm.K % 2) == 0, prm.dY != nullptr);
    if (!h) return hipErrorInvalidValue;
    auto kfn = reinterpret_cast<void (*)(const FusedParams)>(const_cast<void*>(h));
    (void)hipFuncSetAttribute(h, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(kfn, dim3(grid_x, groups), dim3(kWavesPerWG * 64), lds, st, prm);
    return hipGetLastError();
}

struct BwdPlan {
    bool ok;
    int NKT, NW, TPW, NS, HPc, XSC, LD, nch;
    size_t lds;
};

BwdPlan make_bwd_plan(const FusedPlan& pl, int T) {
    BwdPlan bp{};
    if (!pl.ok) return bp;
    bp.NKT = ceil_div(pl.R, 16);
    bp.NW = 16;                                        // 4 waves per SIMD; k-tiles are dealt to them SIMD-balanced in-kernel
    bp.TPW = ceil_div(bp.NKT, bp.NW);
    bp.HPc = 16 * bp.NKT;
    bp.LD = 2 * pl.FP + 16;
    bp.NS = 0;
    for (int ns = 64; ns >= 16; ns >>= 1)
        if (ns * (2 * pl.FP) / 4 <= kDtPF * bp.NW * 64) { bp.NS = ns; break; }
    if (bp.NS == 0 || bp.TPW > 3) return bp;
    bp.XSC = bp.NS + 2 * bp.HPc;
    if (bp.XSC > 2 * bp.NW * 64) return bp;
    bp.lds = (size_t)2 * ((bp.XSC + 3) / 4 * 4 + (size_t)bp.NS * bp.LD) * 4;
    if (bp.lds > (size_t)kMaxLds) return bp;
    bp.nch = ceil_div(T, bp.NS);
    bp.ok = true;
    return bp;
}

hipError_t launch_dtaps(const DtapsParams& prm, int rt, int tpw, int groups, size_t lds, int grid_x, hipStream_t st) {
    const void* h = leaf_inst_dtaps(rt, tpw, (prm.K % 2) == 0);
    if (!h) return hipErrorInvalidValue;
    auto kfn = reinterpret_cast<void (*)(const DtapsParams)>(const_cast<void*>(h));
    (void)hipFuncSetAttribute(h, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(kfn, dim3(grid_x, groups), dim3(prm.NW * 64), lds, st, prm);
    return hipGetLastError();
}

// the fused kernel over all filter groups: full groups of rt_main tiles, then the remainder group
hipError_t launch_fused_groups(FusedParams prm, const FusedPlan& pl, hipStream_t st) {
    const int cus = num_cus();
    const int wg_needed = ceil_div(prm.total_tasks, kWavesPerWG);
    if (pl.groups_main > 0) {
        prm.tile_base = 0;
        const int gx = std::max(1, std::min(wg_needed, std::max(1, cus / pl.groups_main)));
        hipError_t e = launch_fused(prm, pl.rt_main, pl.noff_t, pl.groups_main, fused_lds_bytes(pl.R, pl.rt_main, pl.XS), gx, st);
        if (e != hipSuccess) return e;
    }
    if (pl.rt_rem > 0) {
        prm.tile_base = pl.groups_main * pl.rt_main;
        const int gx = std::max(1, std::min(wg_needed, cus));
        hipError_t e = launch_fused(prm, pl.rt_rem, pl.noff_t, 1, fused_lds_bytes(pl.R, pl.rt_rem, pl.XS), gx, st);
        if (e != hipSuccess) return e;
    }
    return hipSuccess;
}

FusedParams base_fused_params(const FusedPlan& pl, int B, int T, int F, int K, int hop, int tuning_desync) {
    FusedParams prm{};
    prm.B = B; prm.T = T; prm.TP = pl.TP; prm.F = F; prm.FP = pl.FP; prm.K = K; prm.hop = hop; prm.padL = pl.padL;
    prm.KS = pl.KS; prm.Hf = pl.Hf; prm.xshift = pl.xshift; prm.NU = pl.NU; prm.HP = pl.HP; prm.XS = pl.XS;
    prm.q_lo = pl.q_lo; prm.nq = pl.nq; prm.noff = pl.noff; prm.total_tasks = B * pl.nq; prm.GJ = pl.GJ;
    // half a unit of MFMA work is ~ 16*kUB samples x 2*RT tiles x KS k-steps x 32 cycles; s_sleep(127) ~ 8.1k cycles
    prm.desync_sleeps = tuning_desync >= 0 ? tuning_desync
                                           : std::max(1, (int)((long long)kUB * 2 * pl.rt_main * pl.KS * 32 / 2 / 8128));
    return prm;
}

// workspace layout of the fused backward (floats)
struct BwdLayout {
    size_t W, G, Gs, meta, part, raw, ema, gpre, gcols, rowsum, dwpart, dHpart, dY, total;
};

BwdLayout bwd_layout(const FusedPlan& pl, const BwdPlan& bp, int B, int T, int F, int cus) {
    BwdLayout L{};
    size_t o = 0;
    auto take = [&](size_t n) { const size_t at = o; o += align_up(n, 64); return at; };
    L.W = take(pl.w_floats);
    L.G = take(pl.g_floats);
    L.Gs = take(pl.g_floats);
    L.meta = take(pl.meta_ints);
    L.part = take(pl.part_floats);
    L.raw = take((size_t)B * F * pl.TP);
    L.ema = take((size_t)B * F * pl.TP);
    L.gpre = take((size_t)B * F * pl.TP);
    L.gcols = take((size_t)B * pl.TP * pl.FP);
    L.rowsum = take((size_t)B * F * 4);
    L.dwpart = take((size_t)cus * kWavesPerWG * pl.FP);
    L.dHpart = take((size_t)(cus + 1) * 16 * bp.NKT * 2 * pl.FP);     // + 1 slab for the reduced sum
    L.dY = take((size_t)B * T * 2 * pl.FP);
    L.total = o;
    return L;
}

#ifndef LEAF_FFT_FORCE_GENERIC
#define LEAF_FFT_FORCE_GENERIC 0       // measurement only: run the static geometries through the generic-pooling instance
#endif
// ---- FFT (overlap-save) forward plan
struct FftPlan {
    bool ok;
    int L, nblk, NT, GZ, g_bufs, fq, nfq, TP, padL, scr_floats, nslot;
    size_t lds, h_floats, gz_floats, part_floats;
    size_t band_stat, band_dyn;   // tables of the band-limited filter tasks (leaf_band.hpp), 0 where they do not apply
};
// ---- band-limited filter tasks: table layout (float offsets).  The per-filter records and the decimated pooling windows depend
// on the parameters only and sit behind the overlap-save tables (also in the frozen-parameter tables of
// leaf_fft_prepare_tables_f32); the edge tables and the edge list depend on the clip length as well and follow them in a
// forward call's workspace (leaf_forward_prepared_f32: behind the partial sums).
constexpr int kMaxCusForSpec0 = 512;                     // workgroups whose first-block spectrum the table launch computes (>= #CUs)
struct BandLayout {
    size_t rec, gz, stat;          // parameter-only part
    size_t edge, elist, spec0, dyn;   // per-call part (offsets from its own base); spec0: the workgroups' first-block spectra
};
inline BandLayout band_layout(int F, int K, int hop) {
    BandLayout bl{};
    if (!band_geometry_ok(K, hop) || F > kBandMaxFilters) return bl;
    size_t o = 0;
    bl.rec = o; o += align_up((size_t)4 * F, 64);
    bl.gz = o; o += align_up((size_t)F * band_gz_floats(K, hop), 64);
    bl.stat = o;
    o = 0;
    bl.edge = o; o += align_up((size_t)F * 2 * kBandMaxEdge * 512, 64);
    bl.elist = o; o += align_up((size_t)4 * kBandMaxEdge, 64);
    bl.spec0 = o; o += align_up((size_t)kMaxCusForSpec0 * kWgRingFwdFloat2 * 2, 64);
    bl.dyn = o;
    return bl;
}
// Frames reg_lo .. reg_hi take the shift-invariant decimated window (its tails stay inside the clip); every other frame is an
// edge frame with one table per block its (cut) window meets.  false: more edge entries than the tables hold (tiny clips).
inline bool band_edges(int T, int K, int hop, int L, int padL, BandParams& bp, BandEdge (&e)[kBandMaxEdge]) {
    const int TP = (T - 1) / hop + 1, lphi = band_lphi(16);
    int lo = ceil_div(padL + lphi, hop), hi = (T - K - lphi + padL) >= 0 ? (T - K - lphi + padL) / hop : -1;
    hi = std::min(hi, TP - 1);
    if (hi < lo) { lo = TP; hi = TP - 1; }                  // every frame is an edge frame
    bp.reg_lo = lo;
    bp.reg_hi = hi;
    int n = 0;
    for (int m = 0; m < TP; ++m) {
        if (m >= lo && m <= hi) continue;
        const int ws = m * hop - padL;
        for (int c = std::max(0, ws) / L; c * L < std::min(T, ws + K); ++c) {
            const int a = std::max({c * L, 0, ws}), b = std::min({(c + 1) * L, T, ws + K});
            if (a >= b) continue;
            if (n == kBandMaxEdge) return false;
            e[n++] = BandEdge{c, m, a, b};
        }
    }
    bp.n_edge = n;
    return true;
}

FftPlan make_fft_plan(int B, int T, int F, int K, int hop) {
    FftPlan fp{};
    if (K < 2 || K > 64 * kPoolRowsMax - 63) return fp;         // pooling rows per window <= kPoolRowsMax (K <= 1217)
    fp.padL = K / 2 + K % 2 - 1;
    fp.TP = (T - 1) / hop + 1;
    fp.L = fft_block_len(K, hop, fft_static_geometry(K, hop) && !LEAF_FFT_FORCE_GENERIC);
    fp.nblk = ceil_div(T, fp.L);
    fp.NT = ceil_div(K + 63, 64);
    fp.GZ = (kGPad + K + 256 + 3) / 4 * 4;                   // pooling reads run up to 3 rows + 63 lanes past the window
    // filters per task: as many as keeps one wave slot per SIMD pair busy everywhere -- fewer filters per task means the
    // block's forward transform is repeated more often, which only matters once the chip is full
    fp.fq = (int)std::min<long long>(kFftFQ, std::max<long long>(1, (long long)B * fp.nblk * F / ((long long)num_cus() * kFftWaves)));
    fp.nfq = ceil_div(F, fp.fq);
    fp.scr_floats = 32 * 65;                                 // transposes only: the energies stay in registers
    const size_t scr = (size_t)fp.scr_floats;
    for (fp.g_bufs = 2; fp.g_bufs >= 1; --fp.g_bufs) {          // double-buffer the pooling row when LDS allows
        fp.lds = ((size_t)kTwFloats + kFftWaves * (scr + fp.g_bufs * (size_t)fp.GZ)) * 4;
        if (fp.lds <= (size_t)kMaxLds) break;
    }
    if (fp.g_bufs < 1) return fp;
    if ((long long)B * fp.nblk >= (1ll << 30) || F > 65535) return fp;
    fp.h_floats = (size_t)F * kFftN * 2;
    fp.gz_floats = (size_t)F * fp.GZ;
    fp.nslot = (K - 1 > fp.L) ? 3 : 2;                       // blocks a frame's window can meet
    if (K - 1 > 2 * fp.L) return fp;
    fp.part_floats = (size_t)B * fp.TP * fp.nslot * F;
    fp.band_stat = band_layout(F, K, hop).stat;
    fp.band_dyn = band_layout(F, K, hop).dyn;
    fp.ok = true;
    return fp;
}

// ---- 4096-sample plan (leaf_fft_wg4k.hpp): the 32 kHz LEAF geometry.  LEAF_ALGO_FFT_WG means THIS kernel for that geometry
// at every batch size (so that a clip is bit-identical across batch compositions); the frozen-parameter entry points
// (leaf_fft_prepare_tables_f32 / leaf_forward_prepared_f32) keep the 2048-sample tables and kernels.
#ifndef LEAF_FFT_NO_4K
#define LEAF_FFT_NO_4K 0               // measurement only: 1 routes K = 801 through the 2048-sample workgroup kernel
#endif
struct Fft4kPlan {
    bool ok;
    bool generic;          // run-time-geometry kernel (leaf_fft_wgg4k.hpp); false: the static K = 801 / hop = 320 instance
    int L, nblk, nslot, TP, padL, RG, nw;
    size_t lds;
    size_t tab_floats, grow_floats, part_floats;
    size_t band_floats;    // tables of the band tasks (static instance only): records | decimated pooling windows | edge tables | edge list
};
// band tables of the 4096-sample plan (float offsets from their base)
struct Band4kLayout { size_t rec, gz, edge, elist, total; };
inline Band4kLayout band4k_layout(int F, int K, int hop) {
    Band4kLayout bl{};
    size_t o = 0;
    bl.rec = o; o += align_up((size_t)4 * F, 64);
    bl.gz = o; o += align_up((size_t)F * band4k_gz_floats(K, hop), 64);
    bl.edge = o; o += align_up((size_t)F * kBandMaxEdge * 512, 64);
    bl.elist = o; o += align_up((size_t)4 * kBandMaxEdge, 64);
    bl.total = o;
    return bl;
}
// LEAF_NO_4K=1 (environment, tools / tests only): keep every window on the 2048-sample plan
inline bool fft4k_disabled() {
    static const bool off = [] { const char* e = getenv("LEAF_NO_4K"); return e && atoi(e) != 0; }();
    return off || LEAF_FFT_NO_4K || LEAF_FFT_FORCE_GENERIC;
}
FftKernel pick_fft_wgg4k_kernel(int K) { return as_fft_kernel(leaf_inst_fft_wgg4k(fft_wgg4k_taps_per_lane(K))); }
Fft4kPlan make_fft4k_plan(int B, int T, int F, int K, int hop) {
    Fft4kPlan fp{};
    if (fft4k_disabled()) return fp;
    fp.padL = K / 2 + K % 2 - 1;
    fp.TP = (T - 1) / hop + 1;
    if ((long long)B * ceil_div(T, 2048) >= (1ll << 30) || F > 65535) return fp;
    if (K == 801 && hop == 320) {
        fp.L = 3200;
        fp.RG = kWg4RowFloats;
        fp.nw = LEAF_4K_FWD_NW;
        fp.lds = fft_wg4k_lds_bytes(LEAF_4K_FWD_NW);
        if (F <= kBandMaxFilters) fp.band_floats = band4k_layout(F, K, hop).total;
    } else {
        // any other odd window from K = 833 (where the 2048-sample plan drops below half valid outputs) to 2049
        if (!(K & 1) || K < 833 || K > 2049) return fp;
        fp.generic = true;
        fp.L = (kFft4N - K + 1) & ~1;
        if ((fp.L + K - 2) / hop + 2 > kWgg4MaxFrames) return fp;       // frames a block meets: parked in LDS between the halves
        fp.RG = fft_wgg4k_row_floats(K);
        fp.nw = 12;
        const int fbn = fft_wgg4k_frame_floats(K, hop);
        while (fp.nw > 6 && fft_wgg4k_lds_bytes(fp.nw, K, fbn) > (size_t)kMaxLds) --fp.nw;
        fp.lds = fft_wgg4k_lds_bytes(fp.nw, K, fbn);
        if (fp.lds > (size_t)kMaxLds) return fp;
    }
    fp.nblk = ceil_div(T, fp.L);
    fp.nslot = 2;                                                        // K - 1 <= 2048 <= L: a window meets at most two blocks
    fp.tab_floats = (size_t)F * kFft4TabFloats;
    fp.grow_floats = (size_t)F * 2 * fp.RG + kFft4WtFloats;            // + the shared twiddle table w^e of the static kernel's odd half
    fp.part_floats = (size_t)B * fp.TP * fp.nslot * F;
    fp.ok = true;
    return fp;
}
size_t fft4k_workspace_floats(const Fft4kPlan& fp, int B) {
    return align_up(fp.tab_floats, 64) + align_up(fp.grow_floats, 64) + align_up(fp.part_floats, 64) + fp.band_floats + align_up((size_t)B, 64);
}
static_assert(fft_wg4k_bwd_dx_lds_bytes(kWg4BwdDxWaves) <= (size_t)kMaxLds, "LDS budget");
static_assert(fft_wg4k_lds_bytes(12) <= (size_t)kMaxLds && fft_wg4k_bwd_lds_bytes(12) <= (size_t)kMaxLds && fft_wgg4k_lds_bytes(6, 2049, kWgg4MaxFrames) <= (size_t)kMaxLds, "LDS budget");

// ---- which instantiation of leaf_fft_kernel serves a geometry.  Odd K: real-spectrum kernels (the taps are Hermitian
// about the centre tap); even K: complex spectrum.  The backward instances exist for the real-spectrum form only.

// ---- workgroup-per-block variant (leaf_fft_wg.hpp): static odd-window geometries; worth it once every CU gets blocks
struct FftWgLaunch {
    FftKernel fn;
    int nw;
    size_t lds;
    bool fused_finalize = false;   // the kernel deals blocks contiguously and finalizes the clips it owns (FftParams::fin)
    FftKernel fn_stream = nullptr; // the STREAM variant (whole clips per workgroup: finalizes as the blocks complete), if any
    int sk = 0, shop = 0;          // its compile-time geometry (for the LDS size, which grows with F)
    bool lds_sums = false;         // the kernel can keep the per-frame sums of whole clips in LDS (fin_fused = 3)
};
// 12 waves (3 per SIMD, full transposition scratch) by default: with the swap-free cross stage the column-half transposition
// of the 16-wave form (twice the store instructions) costs more than the fourth wave per SIMD brings (cfg1 0.223 vs 0.227 ms,
// cfg3 0.4215 vs 0.428, cfg4 2.03 vs 2.05: tools/bench_configs.py, interleaved).  LEAF_WG_WAVES=16|12 (environment, tools
// only) overrides the choice for A/B measurements.
FftWgLaunch pick_fft_wg_kernel(int K, int hop) {
    if (LEAF_FFT_FORCE_GENERIC) return {nullptr, 0, 0};
    static const int forced = [] { const char* e = tools_env("LEAF_WG_WAVES"); return e ? atoi(e) : 0; }();
    const bool w16 = forced == 16 || forced == 14;
    int nw = 0;
#ifdef LEAF_WG_NW
    if (K == 401 && hop == 160) nw = LEAF_WG_NW;                          // A/B builds only
    else
#endif
    if (K == 401 && hop == 160) nw = w16 ? 16 : 12;
    else if (K == 801 && hop == 320) nw = w16 ? 14 : 10;               // 16 do not fit the LDS
    else if (K == 201 && hop == 80) nw = w16 ? 16 : 12;
    else return {nullptr, 0, 0};
    return {as_fft_kernel(leaf_inst_fft_wg(K, nw, false)), nw, fft_wg_lds_bytes(nw, K), true,
            as_fft_kernel(leaf_inst_fft_wg(K, nw, true)), K, hop, true};
}
// Any other window the 2048-sample plan covers -- odd or even -- takes the run-time-geometry workgroup kernel
// (leaf_fft_wgg.hpp): one instantiation per bucket of taps-per-lane and window parity, as many waves (<= 12: three per
// SIMD's registers) as the LDS holds energy rows for.
FftWgLaunch pick_fft_wgg_kernel(const FftPlan& fp, int K, int hop) {
    (void)hop;
    if (!fp.ok || K < 64 || K > 64 * 19) return {nullptr, 0, 0};          // 19 taps per lane at most
    // the full transposition scratch (fewer LDS store instructions per transform) where the LDS holds it for at least
    // `full_min` = 11 waves (instantiated for the buckets up to 10 taps per lane: 11.025 kHz 0.227 -> 0.209 ms at 12 waves,
    // 22.05 / 24 kHz -1..2 % at 11); LEAF_WGG_FULL=0|12|11 (environment, tools only)
    static const int full_min = [] { const char* e = tools_env("LEAF_WGG_FULL"); return e ? atoi(e) : 11; }();
    const int ni = fft_wgg_taps_per_lane(K);
    if (full_min > 0 && ni <= 10) {
        int nwf = 12;
        while (nwf > full_min && fft_wgg_lds_bytes_full(nwf, K) > (size_t)kMaxLds) --nwf;
        if (fft_wgg_lds_bytes_full(nwf, K) <= (size_t)kMaxLds)
            return {as_fft_kernel(leaf_inst_fft_wgg(ni, false)), nwf, fft_wgg_lds_bytes_full(nwf, K), true};
    }
    int nw = 12;
    while (nw > 6 && fft_wgg_lds_bytes(nw, K) > (size_t)kMaxLds) --nw;
    const size_t lds = fft_wgg_lds_bytes(nw, K);
    if (lds > (size_t)kMaxLds) return {nullptr, 0, 0};
    return {as_fft_kernel(leaf_inst_fft_wgg(ni, true)), nw, lds, true};
}
static_assert(fft_wg_lds_bytes(12, 401) <= (size_t)kMaxLds && fft_wg_lds_bytes(10, 801) <= (size_t)kMaxLds &&
              fft_wg_lds_bytes(16, 401) <= (size_t)kMaxLds && fft_wg_lds_bytes(14, 801) <= (size_t)kMaxLds, "LDS budget");
// AUTO takes the workgroup variant when the batch gives every CU at least one block; below that the per-wave kernel
// (one task per wave, filters-per-task adapted to the batch) has the shorter critical path.
bool fft_wg_available(const FftPlan& fp, int K, int hop) {
    return fp.ok && (pick_fft_wg_kernel(K, hop).fn != nullptr || pick_fft_wgg_kernel(fp, K, hop).fn != nullptr);
}
// Blocks from which the workgroup kernels beat the per-wave kernel: a little under half a block per CU (measured crossover,
// tools/sweep_batch_wg.py: 80 -> 160 blocks at 16 kHz, 60 -> 120 at 22.05 kHz, 80 -> 120 at 8 kHz, 68 -> 136 4096-sample
// blocks at 48 kHz; one block per workgroup costs the same 42 / 53 / 104 us from 1 to 256 blocks)
inline long long fft_wg_min_blocks() { return (long long)num_cus() * 7 / 16; }
// the same question for the backward kernels, in sixteenths of a block per CU (measured, tools/sweep_batch_wg_bwd.py: the static
// kernel against the per-wave backward crosses between 320 and 480 blocks at 16 kHz, the run-time-geometry one between 120
// and 180 at 22.05 kHz, the 4096-sample one between 68 and 136 at 48 kHz); LEAF_WG_BWD_MIN_BLOCKS (environment, tools only)
// overrides it for the sweep
inline long long fft_wg_bwd_min_blocks(int sixteenths) {
    static const long long forced = [] { const char* e = tools_env("LEAF_WG_BWD_MIN_BLOCKS"); return e ? atoll(e) : -1ll; }();
    return forced >= 0 ? forced : (long long)num_cus() * sixteenths / 16;
}
bool fft_wg_auto(const FftPlan& fp, int B, int K, int hop) {
    return fft_wg_available(fp, K, hop) && (long long)B * fp.nblk >= fft_wg_min_blocks() &&
           (K >= 224 || fft_static_geometry(K, hop));
}
FftKernel pick_fft_kernel(const FftPlan& fp, int K, int hop, bool bwd) {
    const bool stat = fft_static_geometry(K, hop) && fp.g_bufs == 2 && !LEAF_FFT_FORCE_GENERIC;
    // odd and even windows alike: real-spectrum kernels (even K: Hermitian K - 1 taps + the unpaired tap in the time domain)
    if (stat && (K & 1)) return as_fft_kernel(leaf_inst_fft(K, 1, 1, bwd ? 1 : 0));
    return as_fft_kernel(leaf_inst_fft(0, fp.g_bufs == 2 ? 1 : 0, (K & 1) ? 1 : 2, bwd ? 1 : 0));
}

// Even K (real-spectrum form): the unpaired taps live behind the real spectra, in the second half of the float2 slab that
// the (retired) complex-spectrum layout sized: [F][2048] floats of spectra, then [F][2] floats.  Odd K: none.
float* fft_lone_taps(float* tables, int F, int K) { return (K & 1) ? nullptr : tables + (size_t)F * kFftN; }

// Floats of the parameter-derived tables of the FFT path: filter spectra, pooling rows, identity column map.
// (`band_dyn`: + the per-call tables of the band-limited filter tasks -- the workspace of a forward call has them, the
// frozen-parameter tables of leaf_fft_prepare_tables_f32 do not: the edge tables depend on the clip length)
size_t fft_table_floats(const FftPlan& fp, int F, bool band_dyn = true) {
    return align_up(fp.h_floats, 64) + align_up(fp.gz_floats, 64) + align_up((size_t)F, 64) + fp.band_stat + (band_dyn ? fp.band_dyn : 0);
}
// (+ B floats behind the partial sums: the per-clip scales of LEAF_FLAG_PEAKNORM)
size_t fft_workspace_floats(const FftPlan& fp, int F, int B) {
    return fft_table_floats(fp, F) + align_up(fp.part_floats, 64) + (LEAF_TRACE ? 16 * 64 * 2 : 0) + align_up((size_t)B, 64);
}

// ---- single-launch small-batch forward (leaf_fft_small.hpp): one workgroup per (clip, filter)
struct SmallPlan {
    bool ok;
    bool split;            // two workgroups of seven waves per (clip, filter) (leaf_fft_small.hpp, SPLIT): while 2 B F workgroups get a CU each
    int nblk, ring, TP;
    size_t lds;
    size_t carry_floats;   // SPLIT: the seam's EMA states, [B][F] (ticket, value) pairs, behind the per-clip scales in the workspace
};
#ifndef LEAF_SMALL_ROUNDS
#define LEAF_SMALL_ROUNDS 2            // rounds of (clip, filter) workgroups over the CUs up to which a batch takes the one-launch kernel (2: 7 .. 12 clips of the default front end 41 -> 30 us; 4: slower than three launches from 16 clips)
#endif
SmallPlan make_small_plan(int B, int T, int F, int K, int hop) {
    SmallPlan sp{};
    if (LEAF_FFT_FORCE_GENERIC || !((K == 401 && hop == 160) || (K == 201 && hop == 80))) return sp;
    const int L = fft_block_len(K, hop, true);
    sp.nblk = ceil_div(T, L);
    sp.TP = (T - 1) / hop + 1;
    sp.ring = std::min(sp.nblk, kSmallRing);
    sp.lds = fft_small_lds_bytes(kSmallWaves, sp.TP);
    // every (clip, filter) pair gets a CU of its own in one round (LEAF_SMALL_ROUNDS = 2: or in two); clips of up to two ring passes
    sp.ok = (long long)B * F <= (long long)LEAF_SMALL_ROUNDS * num_cus() && F <= 65535 && B <= 65535 && sp.nblk <= kSmallMaxBlocks && sp.lds <= (size_t)kMaxLds;
    static const bool split_off = [] { const char* e = tools_env("LEAF_SMALL_SPLIT"); return e && atoi(e) == 0; }();   // tools only: A/B
    if (sp.ok && !split_off && 2ll * B * F <= num_cus() && sp.nblk >= 2 && sp.nblk <= 2 * (kSmallSplitRing - 1) &&
        fft_small_split_lds_bytes(sp.TP) <= (size_t)kMaxLds) {
        sp.split = true;
        sp.lds = fft_small_split_lds_bytes(sp.TP);
        sp.carry_floats = align_up((size_t)B * F * 4, 64);
    }
    return sp;
}

// AUTO: the overlap-save FFT kernel whenever its plan fits and the window is long enough to pay for the transforms --
// its cost per block does not depend on K, the direct MFMA kernel's grows with K.  Measured on MI355X
// (tools/sweep_window.py, B = 256 x 1 s): K = 101/151/201: 338/289/415 us FFT vs 249/317/382 us MFMA (a tie);
// K = 251: 389 vs 841 us; K = 401: 386 vs 1122 us; K = 801: 1.43 vs 4.24 ms.  Batch size does not enter: with the
// filters-per-task adaptation the FFT path also wins at B = 1 (37 vs 58 us, tools/sweep_small_batch.py).
// Short windows / geometries the FFT plan rejects -> MFMA; staged as the last resort.
int auto_algo(int B, int T, int F, int K, int hop, bool allow_small = true) {
    if (allow_small && make_small_plan(B, T, F, K, hop).ok) return LEAF_ALGO_FFT_SMALL;  // a handful of clips: tables, transforms and PCEN in one launch
    const FftPlan fp = make_fft_plan(B, T, F, K, hop);
    const Fft4kPlan f4 = make_fft4k_plan(B, T, F, K, hop);               // long windows: 4096-sample blocks from ~half a block per CU
    if (f4.ok && (long long)B * f4.nblk >= fft_wg_min_blocks()) return LEAF_ALGO_FFT_WG;
    if (f4.ok) {                                                          // ... below that the 2048-sample per-wave kernel, if it fits
        // (also for the static K = 801 geometry: LEAF_ALGO_FFT_WG always means the 4096-sample kernel there, which the
        // threshold above just ruled out)
        if (fp.ok) return LEAF_ALGO_FFT;
        return make_plan(B, T, F, K, hop).ok ? LEAF_ALGO_MFMA : LEAF_ALGO_STAGED;
    }
    if (fft_wg_auto(fp, B, K, hop)) return LEAF_ALGO_FFT_WG;
    if (fp.ok && (K >= 224 || fft_static_geometry(K, hop))) return LEAF_ALGO_FFT;
    return make_plan(B, T, F, K, hop).ok ? LEAF_ALGO_MFMA : LEAF_ALGO_STAGED;
}

inline bool misaligned(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 3u) != 0; }

// Every inst_*.hip was compiled with the parameter-struct layouts this unit has (the handles are opaque, a per-unit
// macro divergence would otherwise be a silent parameter mismatch at launch).  Checked once; a mismatch fails every entry
// point with LEAF_ERR_LAUNCH.
bool inst_layouts_ok() {
    static const bool ok =
        leaf_layout_fft() == leaf_layout_hash_fft() &&
        leaf_layout_fft_small() == leaf_layout_hash_small() &&
        leaf_layout_fft_wg() == leaf_layout_hash_fft() &&
        leaf_layout_fft_wg_bwd() == leaf_layout_hash_fft() &&
        leaf_layout_fft_wg_bwd_dx() == leaf_layout_hash_fft() &&
        leaf_layout_fft_wgg() == leaf_layout_hash_fft() &&
        leaf_layout_fft_wgg4k_bwd() == leaf_layout_hash_fft() &&
        leaf_layout_fft_wgg_bwd() == leaf_layout_hash_fft() &&
        leaf_layout_fft_wgg_bwd_dx() == leaf_layout_hash_fft() &&
        leaf_layout_fused() == leaf_layout_hash_bwd();
    return ok;
}

int check_shape(int B, int T, int F, int K, int hop) {
    if (B < 1 || T < 1 || F < 1 || K < 1 || hop < 1) return LEAF_ERR_BAD_SHAPE;
    if ((long long)B * T >= (1ll << 31)) return LEAF_ERR_BAD_SHAPE;
    if (!inst_layouts_ok()) return LEAF_ERR_LAUNCH;                       // a build whose units disagree about the kernel arguments
    return LEAF_OK;
}

// B == 0 is the EMPTY BATCH, not an error: the reference returns a (0, F, T') tensor for it (frontend.py:78-89 ->
// convolution.py:97 on a zero-size batch) and autograd gives zero parameter gradients.  Entry points that take a batch
// accept it, launch nothing (the backward zero-fills the parameter gradients) and return LEAF_OK; the data pointers of an
// empty batch may be NULL.  Every other extent must still be valid.
inline bool empty_batch(int B, int T, int F, int K, int hop) { return B == 0 && T >= 1 && F >= 1 && K >= 1 && hop >= 1; }

size_t staged_workspace_floats(int B, int T, int F, int K, int hop) {
    const int padL = K / 2 + K % 2 - 1;
    const int TP = (T + (K - 1) - K) / hop + 1;
    (void)padL;
    return align_up((size_t)2 * F * K, 64) + align_up((size_t)F * K, 64) + align_up((size_t)B * 2 * F * T, 64) +
           align_up((size_t)B * F * T, 64) + align_up((size_t)B * F * TP, 64);
}

// the row kernel of the overlap-save paths; `own` = the dealing of the main kernel when that kernel finalized the clips it
// owned outright (OwnedClips{} = none: every row is finalized here)
inline void launch_fft_finalize(const FinParams& fin, int B, const OwnedClips& own, hipStream_t st) {
    const int tiles = ceil_div(B * fin.F, kFinKernelRows);
    if (tiles > 2 * num_cus())
        hipLaunchKernelGGL(fft_finalize_kernel<512>, dim3(tiles), dim3(512), 0, st, fin, B, own);
    else
        hipLaunchKernelGGL(fft_finalize_kernel<1024>, dim3(tiles), dim3(1024), 0, st, fin, B, own);
}
// does the contiguous dealing give every clip to a single workgroup?  (then the row kernel has nothing left to do)
inline bool all_clips_owned(const OwnedClips& own) {
    if (own.nblocks <= 0) return false;
    for (int w = 0; w < own.G; ++w)
        if (own.start(w) % own.nblk != 0) return false;
    return true;
}

#define LEAF_LAUNCH_CHECK()                                  \
    do {                                                     \
        if (hipGetLastError() != hipSuccess) return LEAF_ERR_LAUNCH; \
    } while (0)

}  // namespace

extern "C" {

int leaf_abi_version(void) { return LEAF_ABI_VERSION; }

const char* leaf_status_string(int status) {
    switch (status) {
        case LEAF_OK: return "ok";
        case LEAF_ERR_NULL_POINTER: return "null pointer argument";
        case LEAF_ERR_BAD_SHAPE: return "bad shape (T,F,K,hop must be >= 1, B >= 0 and B*T < 2^31)";
        case LEAF_ERR_WORKSPACE: return "workspace missing or too small (see leaf_workspace_bytes)";
        case LEAF_ERR_BAD_ALGO: return "unknown or inapplicable algorithm selector";
        case LEAF_ERR_LAUNCH: return "HIP kernel launch failed";
        case LEAF_ERR_NO_DEVICE: return "no usable gfx950 device";
        case LEAF_ERR_ALIGNMENT: return "buffer not 4-byte aligned";
        case LEAF_ERR_UNSUPPORTED: return "combination not supported (bfloat16 I/O has no backward / no staged path: use float32 buffers; LEAF_FLAG_PEAKNORM needs an overlap-save path)";
    }
    return "unknown status";
}

int leaf_auto_algo(int B, int T, int F, int K, int hop) {
    if (check_shape(B, T, F, K, hop) != LEAF_OK) return LEAF_ERR_BAD_SHAPE;
    return auto_algo(B, T, F, K, hop);
}

int leaf_fft_plan_info(int B, int T, int F, int K, int hop, int* info) {
    if (!info) return LEAF_ERR_NULL_POINTER;
    if (check_shape(B, T, F, K, hop) != LEAF_OK) return LEAF_ERR_BAD_SHAPE;
    const FftPlan fp = make_fft_plan(B, T, F, K, hop);
    const Fft4kPlan f4 = make_fft4k_plan(B, T, F, K, hop);
    if (f4.ok && auto_algo(B, T, F, K, hop) == LEAF_ALGO_FFT_WG) {        // what AUTO runs is the 4096-sample plan
        info[0] = kFft4N; info[1] = f4.L; info[2] = f4.nblk; info[3] = F; info[4] = 1; info[5] = f4.nslot;
        info[6] = 0; info[7] = (int)f4.lds;
        return LEAF_OK;
    }
    if (!fp.ok) return LEAF_ERR_BAD_ALGO;
    info[0] = kFftN; info[1] = fp.L; info[2] = fp.nblk; info[3] = fp.fq; info[4] = fp.nfq; info[5] = fp.nslot;
    info[6] = fp.g_bufs; info[7] = (int)fp.lds;
    return LEAF_OK;
}

int leaf_num_frames(int T, int K, int hop) {
    if (T < 1 || K < 1 || hop < 1) return LEAF_ERR_BAD_SHAPE;
    const int padL = K / 2 + K % 2 - 1, padR = K / 2;
    return (T + padL + padR - K) / hop + 1;
}

size_t leaf_workspace_bytes(int B, int T, int F, int K, int hop, int algo) {
    if (check_shape(B, T, F, K, hop) != LEAF_OK) return 0;
    const ReserveCus reserve(algo);                          // LEAF_ALGO_RESERVE_CUS(k): AUTO resolves as the forward call will
    algo &= 0xff;
    if (algo == LEAF_ALGO_AUTO) algo = auto_algo(B, T, F, K, hop);
    if (algo == LEAF_ALGO_FFT_SMALL)                          // nothing but the per-clip scales of LEAF_FLAG_PEAKNORM
        return make_small_plan(B, T, F, K, hop).ok ? (align_up((size_t)B, 64) + make_small_plan(B, T, F, K, hop).carry_floats) * 4 : 0;
    const FusedPlan pl = make_plan(B, T, F, K, hop);
    const size_t fused = pl.ok ? (align_up(pl.w_floats, 64) + align_up(pl.g_floats, 64) + align_up(pl.meta_ints, 64) +
                                  align_up(pl.part_floats, 64) + (LEAF_TRACE ? 16 * 64 * 2 : 0)) * 4
                               : 0;
    const size_t staged = staged_workspace_floats(B, T, F, K, hop) * 4;
    if (algo == LEAF_ALGO_FFT || algo == LEAF_ALGO_FFT_WG) {
        const FftPlan fp = make_fft_plan(B, T, F, K, hop);
        const Fft4kPlan f4 = make_fft4k_plan(B, T, F, K, hop);
        if (algo == LEAF_ALGO_FFT_WG && f4.ok) return fft4k_workspace_floats(f4, B) * 4;
        if (algo == LEAF_ALGO_FFT_WG && !fft_wg_available(fp, K, hop)) return 0;
        return fp.ok ? fft_workspace_floats(fp, F, B) * 4 : 0;
    }
    if (algo == LEAF_ALGO_MFMA) return fused;
    if (algo == LEAF_ALGO_STAGED) return staged;
    if (algo == LEAF_ALGO_AUTO) {
        const int a = auto_algo(B, T, F, K, hop);
        if (a == LEAF_ALGO_FFT || a == LEAF_ALGO_FFT_WG) return leaf_workspace_bytes(B, T, F, K, hop, a);
        return a == LEAF_ALGO_MFMA ? fused : staged;
    }
    return 0;
}

int leaf_gabor_taps_f32(const float* kernel, int F, int K, float* taps, void* stream) {
    if (!kernel || !taps) return LEAF_ERR_NULL_POINTER;
    if (F < 1 || K < 1) return LEAF_ERR_BAD_SHAPE;
    hipLaunchKernelGGL(taps_direct_kernel, dim3(ceil_div(F * K, 256)), dim3(256), 0, (hipStream_t)stream, kernel, F, K,
                       gabor_bounds(K), taps);
    LEAF_LAUNCH_CHECK();
    return LEAF_OK;
}

int leaf_lowpass_window_f32(const float* pool_w, int F, int K, float* window, void* stream) {
    if (!pool_w || !window) return LEAF_ERR_NULL_POINTER;
    if (F < 1 || K < 1) return LEAF_ERR_BAD_SHAPE;
    hipLaunchKernelGGL(lowpass_window_kernel, dim3(ceil_div(F * K, 256)), dim3(256), 0, (hipStream_t)stream, pool_w, F, K,
                       window);
    LEAF_LAUNCH_CHECK();
    return LEAF_OK;
}

int leaf_gabor_conv_f32(const float* x, int B, int T, const float* kernel, int F, int K, float* y, void* workspace,
                        size_t workspace_bytes, void* stream) {
    if (!x || !kernel || !y) return LEAF_ERR_NULL_POINTER;
    if (check_shape(B, T, F, K, 1) != LEAF_OK || 2 * F > 65535 || B > 65535) return LEAF_ERR_BAD_SHAPE;
    if (!workspace || workspace_bytes < (size_t)2 * F * K * 4) return LEAF_ERR_WORKSPACE;
    float* taps = static_cast<float*>(workspace);
    int rc = leaf_gabor_taps_f32(kernel, F, K, taps, stream);
    if (rc != LEAF_OK) return rc;
    const int padL = K / 2 + K % 2 - 1;
    hipLaunchKernelGGL(conv_staged_kernel, dim3(ceil_div(T, 256), 2 * F, B), dim3(256), 0, (hipStream_t)stream, x, taps, B,
                       T, 2 * F, K, padL, y);
    LEAF_LAUNCH_CHECK();
    return LEAF_OK;
}

int leaf_squared_modulus_f32(const float* y, int B, int F, int T, float* e, void* stream) {
    if (!y || !e) return LEAF_ERR_NULL_POINTER;
    if (B < 1 || F < 1 || T < 1) return LEAF_ERR_BAD_SHAPE;
    const size_t n = (size_t)B * F * T;
    hipLaunchKernelGGL(sqmod_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, y,
                       (size_t)B * F, T, e);
    LEAF_LAUNCH_CHECK();
    return LEAF_OK;
}

int leaf_gaussian_lowpass_f32(const float* e, int B, int F, int T, const float* pool_w, const float* pool_b, int K,
                              int hop, float* pooled, void* workspace, size_t workspace_bytes, void* stream) {
    if (!e || !pool_w || !pooled) return LEAF_ERR_NULL_POINTER;
    if (check_shape(B, T, F, K, hop) != LEAF_OK || F > 65535 || B > 65535) return LEAF_ERR_BAD_SHAPE;
    if (!workspace || workspace_bytes < (size_t)F * K * 4) return LEAF_ERR_WORKSPACE;
    float* g = static_cast<float*>(workspace);
    int rc = leaf_lowpass_window_f32(pool_w, F, K, g, stream);
    if (rc != LEAF_OK) return rc;
    const int TP = leaf_num_frames(T, K, hop);
    const int padL = K / 2 + K % 2 - 1;
    hipLaunchKernelGGL(pool_staged_kernel, dim3(ceil_div(TP, 64), F, B), dim3(64), 0, (hipStream_t)stream, e, g, pool_b, F,
                       T, TP, K, hop, padL, pooled);
    LEAF_LAUNCH_CHECK();
    return LEAF_OK;
}

int leaf_ema_f32(const float* p, int B, int F, int TP, const float* ema_w, float* ema, void* stream) {
    if (!p || !ema_w || !ema) return LEAF_ERR_NULL_POINTER;
    if (B < 1 || F < 1 || TP < 1) return LEAF_ERR_BAD_SHAPE;
    hipLaunchKernelGGL(pcen_rows_kernel, dim3(ceil_div(B * F, 64)), dim3(64), 0, (hipStream_t)stream, p, B * F, F, TP,
                       (const float*)nullptr, (const float*)nullptr, (const float*)nullptr, ema_w, 0.0f, 0, ema);
    LEAF_LAUNCH_CHECK();
    return LEAF_OK;
}

int leaf_pcen_f32(const float* p, int B, int F, int TP, const float* alpha, const float* delta, const float* root,
                  const float* ema_w, float floor_, float* out, void* stream) {
    if (!p || !alpha || !delta || !root || !ema_w || !out) return LEAF_ERR_NULL_POINTER;
    if (B < 1 || F < 1 || TP < 1) return LEAF_ERR_BAD_SHAPE;
    hipLaunchKernelGGL(pcen_rows_kernel, dim3(ceil_div(B * F, 64)), dim3(64), 0, (hipStream_t)stream, p, B * F, F, TP,
                       alpha, delta, root, ema_w, floor_, 1, out);
    LEAF_LAUNCH_CHECK();
    return LEAF_OK;
}

int leaf_pcen_stream_f32(const float* p, int B, int F, int n, const float* alpha, const float* delta, const float* root,
                         const float* ema_w, float floor_, int log1p_, const float* ema_in, float* ema_out, float* out,
                         void* stream) {
    if (!p || !out) return LEAF_ERR_NULL_POINTER;
    if (alpha && (!delta || !root || !ema_w)) return LEAF_ERR_NULL_POINTER;
    if (B < 1 || F < 1 || n < 1) return LEAF_ERR_BAD_SHAPE;
    FinParams q{};
    q.F = F; q.TP = n; q.alpha = alpha; q.delta = delta; q.root = root; q.ema_w = ema_w; q.floor_ = floor_;
    q.mode = alpha ? 1 : (log1p_ ? 2 : 0);
    q.out = out;
    hipLaunchKernelGGL(pcen_stream_kernel, dim3(ceil_div(B * F, 64)), dim3(64), 0, (hipStream_t)stream, p, B * F, n, q, ema_in,
                       ema_out);
    LEAF_LAUNCH_CHECK();
    return LEAF_OK;
}

// ---- stage backwards: what autograd derives for each reference module called on its own ---------------------

size_t leaf_stage_backward_workspace_bytes(int stage, int B, int T, int F, int K, int hop) {
    if (B < 1 || T < 1 || F < 1) return 0;
    const size_t TP = K >= 1 && hop >= 1 ? (size_t)leaf_num_frames(T, K, hop) : 0;
    switch (stage) {
        case LEAF_STAGE_GABOR_CONV: return (align_up((size_t)2 * F * K, 64) + align_up((size_t)B * 2 * F * K, 64)) * 4;
        case LEAF_STAGE_LOWPASS: return (align_up((size_t)F * K, 64) * 2 + align_up((size_t)F, 64)) * 4;
        case LEAF_STAGE_EMA: return (align_up((size_t)B * F * T, 64) + align_up((size_t)B * F, 64)) * 4;      // T = frames here
        case LEAF_STAGE_PCEN: return (align_up((size_t)B * F * T, 64) + align_up((size_t)B * F * 4, 64)) * 4; // T = frames here
    }
    (void)TP;
    return 0;
}

int leaf_gabor_conv_backward_f32(const float* x, int B, int T, const float* kernel, int F, int K, const float* grad_y,
                                 float* g_kernel, float* g_x, void* workspace, size_t workspace_bytes, void* stream) {
    if (!x || !kernel || !grad_y) return LEAF_ERR_NULL_POINTER;
    if (check_shape(B, T, F, K, 1) != LEAF_OK || 2 * F > 65535 || B > 65535) return LEAF_ERR_BAD_SHAPE;
    if (!workspace || workspace_bytes < leaf_stage_backward_workspace_bytes(LEAF_STAGE_GABOR_CONV, B, T, F, K, 1))
        return LEAF_ERR_WORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    float* taps = static_cast<float*>(workspace);
    float* tpart = taps + align_up((size_t)2 * F * K, 64);
    const int padL = K / 2 + K % 2 - 1;
    int rc = leaf_gabor_taps_f32(kernel, F, K, taps, stream);
    if (rc != LEAF_OK) return rc;
    if (g_kernel) {
        hipLaunchKernelGGL(dtaps_partial_kernel, dim3(ceil_div(K, 128), 2 * F, B), dim3(128), 0, st, grad_y, x, T, 2 * F, K,
                           padL, tpart);
        LEAF_LAUNCH_CHECK();
        hipLaunchKernelGGL(dkernel_kernel, dim3(F), dim3(256), 0, st, tpart, taps, kernel, B, F, K, gabor_bounds(K), g_kernel);
        LEAF_LAUNCH_CHECK();
    }
    if (g_x) {
        hipLaunchKernelGGL(dx_kernel, dim3(ceil_div(T, 256), B), dim3(256), 0, st, grad_y, taps, T, 2 * F, K, padL, g_x);
        LEAF_LAUNCH_CHECK();
    }
    return LEAF_OK;
}

int leaf_squared_modulus_backward_f32(const float* y, const float* grad_e, int B, int F, int T, float* grad_y, void* stream) {
    if (!y || !grad_e || !grad_y) return LEAF_ERR_NULL_POINTER;
    if (B < 1 || F < 1 || T < 1) return LEAF_ERR_BAD_SHAPE;
    const size_t n = (size_t)B * F * T;
    hipLaunchKernelGGL(sqmod_bwd_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, y, grad_e,
                       (size_t)B * F, T, grad_y);
    LEAF_LAUNCH_CHECK();
    return LEAF_OK;
}

int leaf_gaussian_lowpass_backward_f32(const float* e, const float* grad_pooled, int B, int F, int T, const float* pool_w,
                                       int K, int hop, float* g_e, float* g_pool_w, float* g_pool_b, void* workspace,
                                       size_t workspace_bytes, void* stream) {
    if (!e || !grad_pooled || !pool_w) return LEAF_ERR_NULL_POINTER;
    if (check_shape(B, T, F, K, hop) != LEAF_OK || F > 65535 || B > 65535) return LEAF_ERR_BAD_SHAPE;
    if (!workspace || workspace_bytes < leaf_stage_backward_workspace_bytes(LEAF_STAGE_LOWPASS, B, T, F, K, hop))
        return LEAF_ERR_WORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    float* g = static_cast<float*>(workspace);
    float* dg = g + align_up((size_t)F * K, 64);
    float* gw_tmp = dg + align_up((size_t)F * K, 64);
    const int TP = leaf_num_frames(T, K, hop);
    const int padL = K / 2 + K % 2 - 1;
    int rc = leaf_lowpass_window_f32(pool_w, F, K, g, stream);
    if (rc != LEAF_OK) return rc;
    if (g_e) {
        hipLaunchKernelGGL(pool_bwd_de_kernel, dim3(ceil_div(T, 256), F, B), dim3(256), 0, st, g, grad_pooled, F, T, TP, K, hop,
                           padL, g_e);
        LEAF_LAUNCH_CHECK();
    }
    if (g_pool_w || g_pool_b) {
        hipLaunchKernelGGL(pool_bwd_dg_kernel, dim3(ceil_div(K, 128), F), dim3(128), 0, st, e, grad_pooled, B, F, T, TP, K, hop,
                           padL, dg);
        LEAF_LAUNCH_CHECK();
        hipLaunchKernelGGL(param_reduce_kernel, dim3(F), dim3(kParamRedThreads), 0, st, grad_pooled, dg, g, (const float*)nullptr,
                           pool_w, B, F, TP, K, 0, (const float*)nullptr, 0, 0, (const int*)nullptr,
                           g_pool_w ? g_pool_w : gw_tmp, g_pool_b, (float*)nullptr, (float*)nullptr, (float*)nullptr,
                           (float*)nullptr);
        LEAF_LAUNCH_CHECK();
    }
    return LEAF_OK;
}

int leaf_ema_backward_f32(const float* p, const float* grad_ema, int B, int F, int TP, const float* ema_w, float* g_p,
                          float* g_ema_w, void* workspace, size_t workspace_bytes, void* stream) {
    if (!p || !grad_ema || !ema_w || !g_p || !g_ema_w) return LEAF_ERR_NULL_POINTER;
    if (B < 1 || F < 1 || TP < 1) return LEAF_ERR_BAD_SHAPE;
    if (!workspace || workspace_bytes < leaf_stage_backward_workspace_bytes(LEAF_STAGE_EMA, B, TP, F, 1, 1))
        return LEAF_ERR_WORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    float* M = static_cast<float*>(workspace);
    float* rowsum = M + align_up((size_t)B * F * TP, 64);
    hipLaunchKernelGGL(ema_bwd_rows_kernel, dim3(ceil_div(B * F, 64)), dim3(64), 0, st, p, grad_ema, B * F, F, TP, ema_w, M, g_p,
                       rowsum);
    LEAF_LAUNCH_CHECK();
    hipLaunchKernelGGL(rows_to_filter_sum_kernel, dim3(F), dim3(256), 0, st, rowsum, B, F, 1, 1, g_ema_w, (float*)nullptr,
                       (float*)nullptr, (float*)nullptr);
    LEAF_LAUNCH_CHECK();
    return LEAF_OK;
}

int leaf_pcen_backward_f32(const float* p, const float* grad_out, int B, int F, int TP, const float* alpha,
                           const float* delta, const float* root, const float* ema_w, float floor_, float* g_p,
                           float* g_alpha, float* g_delta, float* g_root, float* g_ema_w, void* workspace,
                           size_t workspace_bytes, void* stream) {
    if (!p || !grad_out || !alpha || !delta || !root || !ema_w || !g_p || !g_alpha || !g_delta || !g_root || !g_ema_w)
        return LEAF_ERR_NULL_POINTER;
    if (B < 1 || F < 1 || TP < 1) return LEAF_ERR_BAD_SHAPE;
    if (!workspace || workspace_bytes < leaf_stage_backward_workspace_bytes(LEAF_STAGE_PCEN, B, TP, F, 1, 1))
        return LEAF_ERR_WORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    float* M = static_cast<float*>(workspace);
    float* rowsum = M + align_up((size_t)B * F * TP, 64);
    hipLaunchKernelGGL(pcen_bwd_rows_kernel, dim3(ceil_div(B * F, 64)), dim3(64), 0, st, p, grad_out, B * F, F, TP, alpha, delta,
                       root, ema_w, floor_, 1 | 16, M, g_p, rowsum, (const int*)nullptr, 0, (float*)nullptr);
    LEAF_LAUNCH_CHECK();
    hipLaunchKernelGGL(rows_to_filter_sum_kernel, dim3(F), dim3(256), 0, st, rowsum, B, F, 4, 4, g_alpha, g_delta, g_root,
                       g_ema_w);
    LEAF_LAUNCH_CHECK();
    return LEAF_OK;
}

// The overlap-save forward: (tables) -> main kernel -> finalize.  With tables_ready the tables were produced earlier by
// leaf_fft_prepare_tables_f32 from the same parameters (inference with frozen parameters) and the prep launch is skipped.
static int fft_forward(const FftPlan& fp, const void* x, bool io_bf16, int B, int T, const float* kernel, const float* pool_w,
                       const float* pool_b, const float* alpha, const float* delta, const float* root, const float* ema_w,
                       int F, int K, int hop, int mode, void* out, float* tables, float* part, bool tables_ready,
                       hipStream_t st, hipEvent_t* ev, float* pooled_raw, bool use_wg, const float* clip_scale2 = nullptr,
                       float* band_scratch = nullptr) {
    float2* H = reinterpret_cast<float2*>(tables);
    float* Gz = tables + align_up(fp.h_floats, 64);
    int* col_of = reinterpret_cast<int*>(Gz + align_up(fp.gz_floats, 64));
    if (ev) (void)hipEventRecord(ev[0], st);
    // ---- band-limited filter tasks (leaf_band.hpp): narrow-band filters on 256- / 512-point inverse transforms.  Needs the static
    // workgroup kernel of a geometry they are built for (the training forward takes them too: the pooled tensor it saves for
    // the backward differs from the full-transform one by ~1e-6, the backward recomputes with full transforms).  The tables are
    // built by the prep launch itself (fft_prep_band_kernel); with frozen-parameter tables
    // (tables_ready) the parameter-only part is in them and only the edge tables of this clip length are built here, into
    // band_scratch.  The plan takes band_lds bytes of LDS behind everything else.
    BandParams band{};
    size_t band_lds = 0;
    const float2* spec0 = nullptr;
    if (use_wg && !tl_band_off && (!tables_ready || band_scratch)) {
        static const int band_env = [] { const char* e = tools_env("LEAF_BAND"); return e ? atoi(e) : -1; }();   // tools only: 0 off, 1 / 2 force a class
        static const bool force_generic = [] { const char* e = tools_env("LEAF_WG_GENERIC"); return e && atoi(e) != 0; }();   // tools only
        const FftWgLaunch wl = pick_fft_wg_kernel(K, hop);
        const BandLayout bl = band_layout(F, K, hop);
        BandTabArgs ba{};
        if (bl.stat && wl.fn && wl.nw <= 12 && !force_generic && band_env != 0 && fp.nslot == 2 &&
            wl.lds + band_lds_bytes(F) <= (size_t)kMaxLds && band_edges(T, K, hop, fp.L, fp.padL, band, ba.e)) {
            float* bt = reinterpret_cast<float*>(col_of) + align_up((size_t)F, 64);
            float* dyn = tables_ready ? band_scratch : bt + bl.stat;
            ba.T = T; ba.L = fp.L; ba.hop = hop; ba.padL = fp.padL;
            ba.eps2 = kBandEps2; ba.eta = tl_band_strict ? kBandEta : kBandEtaFree; ba.cross = tl_band_strict ? 0 : 1;   // (leaf_band.hpp: round 6's aliasing bound, windows across Nyquist)
            ba.force = band_env > 0 ? band_env : 0;
            ba.rec = reinterpret_cast<int*>(bt + bl.rec); ba.gz = bt + bl.gz; ba.edge = dyn + bl.edge;
            ba.elist = reinterpret_cast<int*>(dyn + bl.elist); ba.n_edge = band.n_edge;
            ba.edge_only = tables_ready ? 1 : 0;
            band.rec = ba.rec; band.gz = ba.gz; band.edge = ba.edge; band.elist = ba.elist;
            band.bias = pool_b; band.smax = tl_band_strict ? 1.0f : 2.0f;   // the energy bound follows this call's bias (leaf_band.hpp)
            band_lds = band_lds_bytes(F);
            // the main kernel's workgroups' FIRST blocks are transformed by this launch too (waves 1..7 of the workgroups (f, 0),
            // idle while wave 0 transforms the taps): the one forward transform nothing in the main kernel overlaps with (eleven
            // waves waited ~12 k cycles for it).  Other launches of this kernel (leaf_fft_prepare_tables_f32, leaf_band_classes_f32)
            // pass no dynamic LDS and spec0 = NULL.
            static const bool spec0_off = [] { const char* e = tools_env("LEAF_SPEC0"); return e && atoi(e) == 0; }();   // tools only: A/B
            const int main_grid = std::max(1, std::min(B * fp.nblk, num_cus()));
            size_t prep_dyn = 0;
            if (LEAF_WG_SPEC0 && !spec0_off && !tables_ready && main_grid <= kMaxCusForSpec0 && wl.nw <= 12) {
                ba.x = x; ba.io_bf16 = io_bf16 ? 1 : 0; ba.B = B; ba.nblk = fp.nblk; ba.G = main_grid;
                ba.spec0 = reinterpret_cast<float2*>(dyn + bl.spec0);
                spec0 = ba.spec0;
                prep_dyn = (size_t)(kPrepWaves - 1) * kWgScrFloats * 4;
                (void)hipFuncSetAttribute(reinterpret_cast<const void*>(fft_prep_band_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)prep_dyn);
            }
            if (!tables_ready || band.n_edge > 0) {
                hipLaunchKernelGGL(fft_prep_band_kernel, dim3(F, tables_ready ? band.n_edge : 2 + band.n_edge), dim3(kPrepWaves * 64), prep_dyn, st,
                                   kernel, pool_w, F, K, fp.GZ, gabor_bounds(K), H, Gz, col_of, ba);
                LEAF_LAUNCH_CHECK();
            }
        }
    }
    if (!tables_ready && !band.rec) {
        hipLaunchKernelGGL(fft_prep_kernel, dim3(F, 1), dim3(kPrepWaves * 64), 0, st, kernel, pool_w, F, K, fp.GZ,
                           gabor_bounds(K), 1, H, Gz, col_of, fft_lone_taps(tables, F, K));
        LEAF_LAUNCH_CHECK();
    }
    if (ev) (void)hipEventRecord(ev[1], st);
    FftParams q{};
    q.x = x; q.io_bf16 = io_bf16 ? 1 : 0; q.H = H; q.Gz = Gz; q.part = part;
    q.B = B; q.T = T; q.TP = fp.TP; q.F = F; q.K = K; q.hop = hop; q.padL = fp.padL;
    q.L = fp.L; q.nblk = fp.nblk; q.GZ = fp.GZ; q.nslot = fp.nslot; q.g_bufs = fp.g_bufs; q.NT = fp.NT; q.fq = fp.fq; q.nfq = fp.nfq;
    q.scr_floats = fp.scr_floats;
    q.total_tasks = B * fp.nblk * fp.nfq;
    q.rot = K / 2;
    q.lone = fft_lone_taps(tables, F, K);
#if LEAF_TRACE
    q.trace = reinterpret_cast<unsigned long long*>(part + align_up(fp.part_floats, 64));
#endif
    const FinParams fin{part, F, fp.TP, SlotGeom{fp.L, fp.padL, K, hop, T, fp.nslot}, pool_b, alpha, delta, root, ema_w, 1e-12f, mode,
                        out, pooled_raw, clip_scale2};
    OwnedClips own{};                                        // which clips the main kernel finalizes itself (none by default)
    bool all_owned = false;
    if (use_wg) {
        // one persistent workgroup per CU walks its blocks through an LDS task queue (leaf_fft_wg.hpp)
        FftWgLaunch wl = pick_fft_wg_kernel(K, hop);
        static const bool force_generic = [] { const char* e = tools_env("LEAF_WG_GENERIC"); return e && atoi(e) != 0; }();   // tools only
        if (!wl.fn || force_generic) {                        // run-time geometry (any other window, odd or even)
            wl = pick_fft_wgg_kernel(fp, K, hop);
        }
        if (!wl.fn) return LEAF_ERR_BAD_ALGO;
        const int grid = std::max(1, std::min(B * fp.nblk, num_cus()));
        static const bool fin_off = [] { const char* e = tools_env("LEAF_FIN_FUSED"); return e && atoi(e) == 0; }();   // tools only: A/B
        if (wl.fused_finalize && LEAF_WG_TAIL && !LEAF_WG_STRIDED && !fin_off) {
            // clip-resident finalize: blocks are dealt contiguously.  Whole clips per workgroup (the batch a multiple of the
            // grid, or fewer clips than CUs never happens here: grid = min(blocks, CUs)) -> the STREAM variant: frame sums in
            // an LDS ring, finalized as the blocks complete, no `part`, no second kernel.  Otherwise a workgroup finalizes the
            // clips it owns outright in its tail and the row kernel below sees the clips that straddle two workgroups.
            own = OwnedClips{B * fp.nblk, grid, fp.nblk};
            all_owned = all_clips_owned(own);
            q.fin = fin;
            q.fin_fused = 1;
            // every workgroup gets the same number of whole clips and their per-frame sums fit behind its scratch: the sums stay
            // in LDS (ds_add_f32 of the two blocks a window meets; the tail reads them there) -- no `part` traffic at all
            static const bool lds_sums_off = [] { const char* e = tools_env("LEAF_LDS_SUMS"); return e && atoi(e) == 0; }();   // tools only: A/B
            if (all_owned && wl.lds_sums && fp.nslot == 2 && !lds_sums_off && !tl_stream_finalize && (B * fp.nblk) % grid == 0) {
                const size_t extra = (((size_t)(B / grid) * F * fp.TP + 7) / 8 * 8 + (size_t)F * 8) * 4;   // the sums + the filters' finalize coefficients
                if (B % grid == 0 && wl.lds + extra + band_lds <= (size_t)kMaxLds) {
                    wl.lds += extra;
                    q.fin_fused = 3;
                }
            }
            static const int stream_env = [] { const char* e = tools_env("LEAF_WG_STREAM"); return e ? atoi(e) : -1; }();   // tools only: A/B
            // LEAF_ALGO_STREAM_FINALIZE asks for it; without the flag it is what runs wherever the workgroups own whole clips
            // but their frame sums do NOT fit the LDS (several clips per workgroup, long clips: BASELINE configs[3], [4]) --
            // there the alternative is the round trip of every partial sum through `part` in HBM (2x the algorithmic traffic)
#ifndef LEAF_STREAM_AUTO
#define LEAF_STREAM_AUTO 1         // 0 (A/B builds, tools/compare_builds.py): streaming finalize only on request, as in round 3
#endif
            const bool want_stream = stream_env >= 0 ? stream_env != 0 : (tl_stream_finalize || (LEAF_STREAM_AUTO && q.fin_fused != 3));
            if (all_owned && wl.fn_stream && fp.nslot == 2 && want_stream) {
                // the longest ring the LDS holds, up to four times the minimum (lag >= 4: the forward tasks never wait)
                int ring = 0;
                for (int r = 4 * wg_stream_ring_min(wl.sk, wl.shop); r >= wg_stream_ring_min(wl.sk, wl.shop); r >>= 1)
                    if (fft_wg_stream_lds_bytes(wl.nw, wl.sk, r, F) + band_lds <= (size_t)kMaxLds) { ring = r; break; }
                if (ring) {
                    wl.fn = wl.fn_stream;
                    wl.lds = fft_wg_stream_lds_bytes(wl.nw, wl.sk, ring, F);
                    q.stream_ring = ring;
                    q.fin_fused = 2;
                }
            }
        }
        if (band.rec) {                                       // (decided before the prep launch, which built the tables)
            q.spec0 = grid == std::max(1, std::min(B * fp.nblk, num_cus())) ? spec0 : nullptr;
            q.band = band;
            q.band.lds_off = (int)(wl.lds / 4);
            wl.lds += band_lds;
        }
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(wl.fn), hipFuncAttributeMaxDynamicSharedMemorySize, (int)wl.lds);
        hipLaunchKernelGGL(wl.fn, dim3(grid), dim3(wl.nw * 64), wl.lds, st, q);
    } else {
        FftKernel kfn = pick_fft_kernel(fp, K, hop, false);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kfn), hipFuncAttributeMaxDynamicSharedMemorySize, (int)fp.lds);
        hipLaunchKernelGGL(kfn, dim3(std::max(1, std::min(ceil_div(q.total_tasks, kFftWaves), num_cus()))), dim3(kFftWaves * 64),
                           fp.lds, st, q);
    }
    LEAF_LAUNCH_CHECK();
    if (ev) (void)hipEventRecord(ev[2], st);
    if (!all_owned) launch_fft_finalize(fin, B, own, st);
    LEAF_LAUNCH_CHECK();
    if (ev) (void)hipEventRecord(ev[3], st);
    return LEAF_OK;
}

int leaf_peak_normalize_f32(const float* x, int B, int T, float* out, void* stream) {
    if (!x || !out) return LEAF_ERR_NULL_POINTER;
    if (B < 1 || T < 1) return LEAF_ERR_BAD_SHAPE;
    hipLaunchKernelGGL(peak_normalize_kernel, dim3(B), dim3(1024), 0, (hipStream_t)stream, x, T, out);
    LEAF_LAUNCH_CHECK();
    return LEAF_OK;
}

static int forward_impl(const void* x, int B, int T, const float* kernel, const float* pool_w, const float* pool_b,
                        const float* alpha, const float* delta, const float* root, const float* ema_w, int F, int K, int hop,
                        int flags, int algo, void* out, void* workspace, size_t workspace_bytes, void* stream,
                        hipEvent_t* ev, float* pooled_raw = nullptr) {
    {
        // the selector, the flag combination and the build's layout check hold for the empty batch too (a bad selector or a
        // mismatched build must not go unnoticed on an empty shard): only then (0, F, T') -- nothing to compute, nothing launched
        const int sel = algo & 0xff;
        if (sel != LEAF_ALGO_AUTO && sel != LEAF_ALGO_STAGED && sel != LEAF_ALGO_MFMA && sel != LEAF_ALGO_FFT && sel != LEAF_ALGO_FFT_WG &&
            sel != LEAF_ALGO_FFT_SMALL)
            return LEAF_ERR_BAD_ALGO;
        if ((flags & LEAF_FLAG_IO_BF16) && sel == LEAF_ALGO_STAGED) return LEAF_ERR_UNSUPPORTED;
        if (B == 0 && !inst_layouts_ok()) return LEAF_ERR_LAUNCH;
    }
    if (empty_batch(B, T, F, K, hop)) return LEAF_OK;
    if (!x || !kernel || !pool_w || !pool_b || !out) return LEAF_ERR_NULL_POINTER;
    const bool use_pcen = (flags & LEAF_FLAG_PCEN) != 0;
    if (use_pcen && (!alpha || !delta || !root || !ema_w)) return LEAF_ERR_NULL_POINTER;
    int rc = check_shape(B, T, F, K, hop);
    if (rc != LEAF_OK) return rc;
    {
        const uintptr_t io_mask = (flags & LEAF_FLAG_IO_BF16) ? 1u : 3u;
        if ((reinterpret_cast<uintptr_t>(x) & io_mask) || (reinterpret_cast<uintptr_t>(out) & io_mask) || misaligned(workspace))
            return LEAF_ERR_ALIGNMENT;
    }
    const int tuning_desync = ((algo >> 8) & 0xff) - 1;      // LEAF_ALGO_TUNE_DESYNC(n); -1 = automatic
    const ReserveCus reserve(algo);                          // LEAF_ALGO_RESERVE_CUS(k): grids of this call leave k CUs free
    algo &= 0xff;
    if (algo != LEAF_ALGO_AUTO && algo != LEAF_ALGO_STAGED && algo != LEAF_ALGO_MFMA && algo != LEAF_ALGO_FFT &&
        algo != LEAF_ALGO_FFT_WG && algo != LEAF_ALGO_FFT_SMALL)
        return LEAF_ERR_BAD_ALGO;
    const FusedPlan pl = make_plan(B, T, F, K, hop);
    const bool io_bf16 = (flags & LEAF_FLAG_IO_BF16) != 0;
    if (io_bf16 && algo == LEAF_ALGO_STAGED) return LEAF_ERR_UNSUPPORTED;             // bf16 I/O is a fused-path feature
    if (algo == LEAF_ALGO_MFMA && !pl.ok) return LEAF_ERR_BAD_ALGO;
    if (algo == LEAF_ALGO_AUTO) algo = auto_algo(B, T, F, K, hop);
    if (io_bf16 && algo == LEAF_ALGO_STAGED) return LEAF_ERR_UNSUPPORTED;
    const size_t need = leaf_workspace_bytes(B, T, F, K, hop, algo);
    if (!workspace || workspace_bytes < need) return LEAF_ERR_WORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    const int mode = (use_pcen ? 1 : 0) | ((flags & LEAF_FLAG_LOG1P) && !use_pcen ? 2 : 0) | (io_bf16 ? 4 : 0);
    const int TP = pl.TP;
    float* ws = static_cast<float*>(workspace);
    // LEAF_FLAG_PEAKNORM: per-clip scales into the tail of the workspace (the last align_up(B, 64) floats the overlap-save
    // plans reserve), applied to the pooled energies by the finalize step
    float* clip_scale2 = nullptr;
    if (flags & LEAF_FLAG_PEAKNORM) {
        if (algo != LEAF_ALGO_FFT && algo != LEAF_ALGO_FFT_WG && algo != LEAF_ALGO_FFT_SMALL) return LEAF_ERR_UNSUPPORTED;
        clip_scale2 = ws + need / 4 - align_up((size_t)B, 64);
        hipLaunchKernelGGL(peak_scale2_kernel, dim3(B), dim3(1024), 0, st, x, io_bf16 ? 1 : 0, T, clip_scale2);
        LEAF_LAUNCH_CHECK();
    }

    if (algo == LEAF_ALGO_FFT_SMALL) {
        // one launch: grid (F, B), every workgroup builds its filter's tables, transforms its clip's blocks and finalizes its row
        const SmallPlan sp = make_small_plan(B, T, F, K, hop);
        if (!sp.ok) return LEAF_ERR_BAD_ALGO;
        using SmallKernel = void (*)(const SmallParams);
        const SmallKernel kfn = reinterpret_cast<SmallKernel>(const_cast<void*>(leaf_inst_fft_small(K, sp.split)));
        if (!kfn) return LEAF_ERR_BAD_ALGO;
        SmallParams q{};
        q.x = x; q.io_bf16 = io_bf16 ? 1 : 0; q.kernel = kernel; q.pool_w = pool_w; q.bd = gabor_bounds(K);
        q.B = B; q.T = T; q.TP = sp.TP; q.F = F; q.nblk = sp.nblk; q.ring = sp.split ? kSmallSplitRing : sp.ring;
        q.fin = FinParams{nullptr, F, sp.TP, SlotGeom{fft_block_len(K, hop, true), K / 2 + K % 2 - 1, K, hop, T, 2}, pool_b, alpha, delta,
                          root, ema_w, 1e-12f, mode, out, pooled_raw, clip_scale2};
        if (ev) { (void)hipEventRecord(ev[0], st); (void)hipEventRecord(ev[1], st); }
        // (the attribute is per function and device: set on every call like the other kernels -- a host-side table lookup)
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kfn), hipFuncAttributeMaxDynamicSharedMemorySize, (int)sp.lds);
        if (sp.split) {
            // this launch's ticket for the seam hand-over (64 bits from a random seed: stale or uninitialised workspace never matches)
            static std::atomic<unsigned long long> ticket{[] { std::random_device rd; return ((unsigned long long)rd() << 32) ^ rd(); }()};
            q.epoch = ticket.fetch_add(1, std::memory_order_relaxed) + 1;
            if (q.epoch == 0) q.epoch = ticket.fetch_add(1, std::memory_order_relaxed) + 1;   // 0 = "taken out" (the consumer resets the slot)
            q.carry = reinterpret_cast<unsigned long long*>(ws);          // (the per-clip scales sit at the workspace's end)
        }
        hipLaunchKernelGGL(kfn, dim3(F, B, sp.split ? 2 : 1), dim3((sp.split ? kSmallSplitWaves : kSmallWaves) * 64), sp.lds, st, q);
        LEAF_LAUNCH_CHECK();
        if (ev) { (void)hipEventRecord(ev[2], st); (void)hipEventRecord(ev[3], st); }
        return LEAF_OK;
    }
    if (algo == LEAF_ALGO_FFT_WG) {
        const Fft4kPlan f4 = make_fft4k_plan(B, T, F, K, hop);
        if (f4.ok) {
            // 4096-sample blocks: tables -> workgroup kernel -> the same finalize kernel (partials keep their layout)
            float* tab = ws;
            float* Grow = tab + align_up(f4.tab_floats, 64);
            float* part = Grow + align_up(f4.grow_floats, 64);
            if (ev) (void)hipEventRecord(ev[0], st);
            float2* Wt = reinterpret_cast<float2*>(Grow + (size_t)F * 2 * f4.RG);
            // band-limited filter tasks (leaf_band.hpp): the static instance runs the filters whose spectrum sits in a 512-bin window
            // of the 4096-point spectrum four to a task on 512-point transforms (decided per call by the prep kernel from the
            // spectrum it has just built; tables by fft4k_band_tab_kernel); LEAF_ALGO_FULL_TRANSFORMS switches them off
            BandParams band{};
            BandTabArgs ba{};
            size_t band_lds = 0;
            static const int band_env = [] { const char* e = tools_env("LEAF_BAND"); return e ? atoi(e) : -1; }();   // tools only: 0 off, 2 every filter
            if (!f4.generic && f4.band_floats && !tl_band_off && band_env != 0 && f4.lds + band_lds_bytes(F) <= (size_t)kMaxLds &&
                band_edges(T, K, hop, f4.L, f4.padL, band, ba.e)) {
                const Band4kLayout bl = band4k_layout(F, K, hop);
                float* bt = part + align_up(f4.part_floats, 64);
                ba.T = T; ba.L = f4.L; ba.hop = hop; ba.padL = f4.padL;
                ba.eps2 = kBandEps2; ba.eta = tl_band_strict ? kBandEta : kBandEtaFree; ba.force = band_env > 0 ? band_env : 0;
                ba.rec = reinterpret_cast<int*>(bt + bl.rec); ba.gz = bt + bl.gz; ba.edge = bt + bl.edge;
                ba.elist = reinterpret_cast<int*>(bt + bl.elist); ba.n_edge = band.n_edge;
                band.rec = ba.rec; band.gz = ba.gz; band.edge = ba.edge; band.elist = ba.elist;
                band.bias = pool_b; band.smax = tl_band_strict ? 1.0f : 2.0f;   // the energy bound follows this call's bias
                band_lds = band_lds_bytes(F);
            }
            hipLaunchKernelGGL(fft4k_prep_kernel, dim3(F), dim3(kPrepWaves * 64), 0, st, kernel, pool_w, F, K, gabor_bounds(K), tab,
                               Grow, f4.RG, Wt, ba);
            LEAF_LAUNCH_CHECK();
            if (band.rec) hipLaunchKernelGGL(fft4k_band_tab_kernel, dim3(F, 1 + band.n_edge), dim3(kPrepWaves * 64), 0, st, pool_w, F, K, ba);
            LEAF_LAUNCH_CHECK();
            if (ev) (void)hipEventRecord(ev[1], st);
            FftParams q{};
            q.x = x; q.io_bf16 = io_bf16 ? 1 : 0; q.H = reinterpret_cast<const float2*>(tab); q.Gz = Grow; q.part = part;
            q.B = B; q.T = T; q.TP = f4.TP; q.F = F; q.K = K; q.hop = hop; q.padL = f4.padL; q.L = f4.L; q.nblk = f4.nblk;
            q.nslot = f4.nslot; q.GZ = f4.RG; q.NT = f4.generic ? fft_wgg4k_frame_floats(K, hop) : 0;
            q.lone = reinterpret_cast<const float*>(Wt);                  // (static 32 kHz kernel: the shared twiddle table travels in `lone`)
            if (band.rec) {
                q.band = band;
                q.band.lds_off = (int)(f4.lds / 4);
            }
            FftKernel kfn = f4.generic ? pick_fft_wgg4k_kernel(K) : as_fft_kernel(leaf_inst_fft_wg4k());
            const size_t lds = f4.lds + band_lds;
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kfn), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            // blocks dealt contiguously; a workgroup finalizes the clips it ran every block of in its tail (as fft_forward)
            const int grid = std::max(1, std::min(B * f4.nblk, num_cus()));
            const FinParams fin{part, F, f4.TP, SlotGeom{f4.L, f4.padL, K, hop, T, f4.nslot}, pool_b, alpha, delta, root, ema_w,
                                1e-12f, mode, out, pooled_raw, clip_scale2};
            static const bool fin_off = [] { const char* e = tools_env("LEAF_FIN_FUSED"); return e && atoi(e) == 0; }();   // tools only: A/B
            OwnedClips own{};
            if (!fin_off) {
                own = OwnedClips{B * f4.nblk, grid, f4.nblk};
                q.fin = fin;
                q.fin_fused = 1;
            }
            hipLaunchKernelGGL(kfn, dim3(grid), dim3(f4.nw * 64), lds, st, q);
            LEAF_LAUNCH_CHECK();
            if (ev) (void)hipEventRecord(ev[2], st);
            if (fin_off || !all_clips_owned(own)) launch_fft_finalize(fin, B, own, st);
            LEAF_LAUNCH_CHECK();
            if (ev) (void)hipEventRecord(ev[3], st);
            return LEAF_OK;
        }
    }
    if (algo == LEAF_ALGO_FFT || algo == LEAF_ALGO_FFT_WG) {
        const FftPlan fp = make_fft_plan(B, T, F, K, hop);
        if (!fp.ok) return LEAF_ERR_BAD_ALGO;
        float* tables = ws;                                        // [spectra | pooling rows | col_of], then the partials
        float* part = ws + fft_table_floats(fp, F);
        return fft_forward(fp, x, io_bf16, B, T, kernel, pool_w, pool_b, alpha, delta, root, ema_w, F, K, hop, mode, out,
                           tables, part, /*tables_ready=*/false, st, ev, pooled_raw, algo == LEAF_ALGO_FFT_WG, clip_scale2);
    }

    if (algo == LEAF_ALGO_MFMA) {
        float* W = ws;
        float* G = ws + align_up(pl.w_floats, 64);
        int* meta = reinterpret_cast<int*>(G + align_up(pl.g_floats, 64));
        int* perm = meta;
        int* col_of = meta + pl.FP;
        int* tile_ks = meta + 2 * pl.FP;
        float* part = G + align_up(pl.g_floats, 64) + align_up(pl.meta_ints, 64);
        if (ev) (void)hipEventRecord(ev[0], st);
        hipLaunchKernelGGL(fused_prep_kernel, dim3(ceil_div(pl.R * 2 * pl.FP + pl.FP * pl.GJ, 256)), dim3(256), 0, st,
                           kernel, pool_w, F, pl.FP, K, pl.R, pl.GJ, gabor_bounds(K), W, G, (float*)nullptr, perm, col_of,
                           tile_ks);
        LEAF_LAUNCH_CHECK();
        if (ev) (void)hipEventRecord(ev[1], st);
        FusedParams prm = base_fused_params(pl, B, T, F, K, hop, tuning_desync);
        prm.x = x; prm.io_bf16 = io_bf16 ? 1 : 0; prm.W = W; prm.G = G; prm.tile_ks = tile_ks; prm.part = part;
#if LEAF_TRACE
        prm.trace = reinterpret_cast<unsigned long long*>(part + align_up(pl.part_floats, 64));
#endif
        if (launch_fused_groups(prm, pl, st) != hipSuccess) return LEAF_ERR_LAUNCH;
        if (ev) (void)hipEventRecord(ev[2], st);
        hipLaunchKernelGGL(finalize_kernel, dim3(B, ceil_div(F, kFinGroup)), dim3(kFinGroup * 64), (size_t)kFinGroup * (kFinStride + 3) * 4, st, part, F, pl.FP, TP, pl.noff,
                           pl.q_lo, pl.q_hi, SlotGeom{}, col_of, pool_b, alpha, delta, root, ema_w, 1e-12f, mode, out, pooled_raw);
        LEAF_LAUNCH_CHECK();
        if (ev) (void)hipEventRecord(ev[3], st);
        return LEAF_OK;
    }

    // staged path: every intermediate of the reference graph is materialised in the workspace
    if (2 * F > 65535 || B > 65535) return LEAF_ERR_BAD_SHAPE;
    const float* xf32 = static_cast<const float*>(x);
    float* outf32 = static_cast<float*>(out);
    float* taps = ws;
    float* g = taps + align_up((size_t)2 * F * K, 64);
    float* y = g + align_up((size_t)F * K, 64);
    float* e = y + align_up((size_t)B * 2 * F * T, 64);
    float* pooled = e + align_up((size_t)B * F * T, 64);
    rc = leaf_gabor_conv_f32(xf32, B, T, kernel, F, K, y, taps, (size_t)2 * F * K * 4, stream);
    if (rc != LEAF_OK) return rc;
    rc = leaf_squared_modulus_f32(y, B, F, T, e, stream);
    if (rc != LEAF_OK) return rc;
    rc = leaf_gaussian_lowpass_f32(e, B, F, T, pool_w, pool_b, K, hop, pooled, g, (size_t)F * K * 4, stream);
    if (rc != LEAF_OK) return rc;
    const size_t n = (size_t)B * F * TP;
    if (pooled_raw && hipMemcpyAsync(pooled_raw, pooled, n * 4, hipMemcpyDeviceToDevice, st) != hipSuccess)
        return LEAF_ERR_LAUNCH;
    if (use_pcen) {
        // floor in place, then PCEN
        hipLaunchKernelGGL(floor_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, pooled, n, 0, pooled);
        LEAF_LAUNCH_CHECK();
        return leaf_pcen_f32(pooled, B, F, TP, alpha, delta, root, ema_w, 1e-12f, outf32, stream);
    }
    hipLaunchKernelGGL(floor_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, pooled, n, mode, outf32);
    LEAF_LAUNCH_CHECK();
    return LEAF_OK;
}

int leaf_forward_f32(const float* x, int B, int T, const float* kernel, const float* pool_w, const float* pool_b,
                     const float* alpha, const float* delta, const float* root, const float* ema_w, int F, int K, int hop,
                     int flags, int algo, float* out, void* workspace, size_t workspace_bytes, void* stream) {
    return forward_impl(x, B, T, kernel, pool_w, pool_b, alpha, delta, root, ema_w, F, K, hop, flags, algo, out, workspace,
                        workspace_bytes, stream, nullptr);
}

int leaf_forward_save_f32(const float* x, int B, int T, const float* kernel, const float* pool_w, const float* pool_b,
                          const float* alpha, const float* delta, const float* root, const float* ema_w, int F, int K, int hop,
                          int flags, int algo, float* out, float* pooled_raw, void* workspace, size_t workspace_bytes,
                          void* stream) {
    if (!pooled_raw) return LEAF_ERR_NULL_POINTER;
    if (flags & LEAF_FLAG_IO_BF16) return LEAF_ERR_UNSUPPORTED;          // the backward is fp32-only
    // forward-only: pooled_raw would be that of the normalised clips while leaf_backward_f32 differentiates against x
    if (flags & LEAF_FLAG_PEAKNORM) return LEAF_ERR_UNSUPPORTED;
    return forward_impl(x, B, T, kernel, pool_w, pool_b, alpha, delta, root, ema_w, F, K, hop, flags, algo, out, workspace,
                        workspace_bytes, stream, nullptr, pooled_raw);
}

int leaf_forward_profiled_f32(const float* x, int B, int T, const float* kernel, const float* pool_w, const float* pool_b,
                              const float* alpha, const float* delta, const float* root, const float* ema_w, int F, int K,
                              int hop, int flags, int algo, float* out, void* workspace, size_t workspace_bytes,
                              void* stream, float* stage_ms) {
    if (!stage_ms) return LEAF_ERR_NULL_POINTER;
    if (empty_batch(B, T, F, K, hop)) { stage_ms[0] = stage_ms[1] = stage_ms[2] = 0.f; return LEAF_OK; }
    hipEvent_t ev[4];
    for (int i = 0; i < 4; ++i)
        if (hipEventCreate(&ev[i]) != hipSuccess) return LEAF_ERR_LAUNCH;
    {
        const ReserveCus reserve(algo);                      // AUTO resolves as the forward call will
        int sel = algo & 0xff;
        if (sel == LEAF_ALGO_AUTO) sel = auto_algo(B, T, F, K, hop);
        if (sel != LEAF_ALGO_MFMA && sel != LEAF_ALGO_FFT && sel != LEAF_ALGO_FFT_WG && sel != LEAF_ALGO_FFT_SMALL) {
            for (int i = 0; i < 4; ++i) (void)hipEventDestroy(ev[i]);
            return LEAF_ERR_BAD_ALGO;
        }
        algo = (algo & ~0xff) | sel;
    }
    int rc = forward_impl(x, B, T, kernel, pool_w, pool_b, alpha, delta, root, ema_w, F, K, hop, flags, algo, out, workspace,
                          workspace_bytes, stream, ev);
    if (rc == LEAF_OK) {
        if (hipEventSynchronize(ev[3]) != hipSuccess) rc = LEAF_ERR_LAUNCH;
        for (int i = 0; i < 3 && rc == LEAF_OK; ++i)
            if (hipEventElapsedTime(&stage_ms[i], ev[i], ev[i + 1]) != hipSuccess) rc = LEAF_ERR_LAUNCH;
    }
    for (int i = 0; i < 4; ++i) (void)hipEventDestroy(ev[i]);
    return rc;
}


// ---- inference with frozen parameters: parameter-derived tables prepared once, reused by every forward
size_t leaf_fft_tables_bytes(int F, int K, int hop) {
    if (F < 1 || K < 1 || hop < 1) return 0;
    const FftPlan fp = make_fft_plan(1, std::max(K, 2 * kFftN), F, K, hop);      // table sizes depend on (F, K) only
    return fp.ok ? fft_table_floats(fp, F, false) * 4 : 0;
}

int leaf_fft_prepare_tables_f32(const float* kernel, const float* pool_w, int F, int K, int hop, void* tables,
                                size_t tables_bytes, void* stream) {
    if (!kernel || !pool_w || !tables) return LEAF_ERR_NULL_POINTER;
    const size_t need = leaf_fft_tables_bytes(F, K, hop);
    if (need == 0) return LEAF_ERR_BAD_ALGO;
    if (tables_bytes < need) return LEAF_ERR_WORKSPACE;
    if (misaligned(tables)) return LEAF_ERR_ALIGNMENT;
    const FftPlan fp = make_fft_plan(1, std::max(K, 2 * kFftN), F, K, hop);
    float* t = static_cast<float*>(tables);
    float* Gz = t + align_up(fp.h_floats, 64);
    int* col_of = reinterpret_cast<int*>(Gz + align_up(fp.gz_floats, 64));
    const BandLayout bl = band_layout(F, K, hop);
    if (bl.stat) {
        // + the parameter-only tables of the band-limited filter tasks (per-filter records, decimated pooling windows)
        float* bt = reinterpret_cast<float*>(col_of) + align_up((size_t)F, 64);
        BandTabArgs ba{};
        ba.hop = hop; ba.padL = fp.padL; ba.L = fp.L; ba.eps2 = kBandEps2; ba.eta = kBandEtaFree; ba.cross = 1;   // (the forward's default rule)
        ba.rec = reinterpret_cast<int*>(bt + bl.rec); ba.gz = bt + bl.gz;
        hipLaunchKernelGGL(fft_prep_band_kernel, dim3(F, 2), dim3(kPrepWaves * 64), 0, (hipStream_t)stream, kernel, pool_w, F, K, fp.GZ,
                           gabor_bounds(K), reinterpret_cast<float2*>(t), Gz, col_of, ba);
    } else {
        hipLaunchKernelGGL(fft_prep_kernel, dim3(F, 1), dim3(kPrepWaves * 64), 0, (hipStream_t)stream, kernel, pool_w, F, K, fp.GZ,
                           gabor_bounds(K), 1, reinterpret_cast<float2*>(t), Gz, col_of, fft_lone_taps(t, F, K));
    }
    LEAF_LAUNCH_CHECK();
    return LEAF_OK;
}

int leaf_band_classes_f32(const float* kernel, const float* pool_w, const float* pool_b, int F, int K, int hop, int* classes,
                          void* workspace, size_t workspace_bytes, void* stream) {
    if (!kernel || !pool_w || !classes) return LEAF_ERR_NULL_POINTER;
    if (F < 1 || K < 1 || hop < 1) return LEAF_ERR_BAD_SHAPE;
    {
        // the 4096-sample plan (K = 801 / hop = 320): one class, decided by fft4k_prep_kernel; classes[f] = 512 or 4096
        const Fft4kPlan f4 = make_fft4k_plan(1, 2 * kFft4N, F, K, hop);
        if (f4.ok && !f4.generic && f4.band_floats) {
            if (!workspace || workspace_bytes < fft4k_workspace_floats(f4, 1) * 4) return LEAF_ERR_WORKSPACE;
            if (misaligned(workspace) || misaligned(classes)) return LEAF_ERR_ALIGNMENT;
            float* tab = static_cast<float*>(workspace);
            float* Grow = tab + align_up(f4.tab_floats, 64);
            float* bt = Grow + align_up(f4.grow_floats, 64) + align_up(f4.part_floats, 64);
            const Band4kLayout b4 = band4k_layout(F, K, hop);
            BandTabArgs ba{};
            ba.hop = hop; ba.padL = f4.padL; ba.L = f4.L; ba.eps2 = kBandEps2; ba.eta = pool_b ? kBandEtaFree : kBandEta;   // (pool_b = NULL: round 5's decision)
            ba.rec = reinterpret_cast<int*>(bt + b4.rec); ba.gz = bt + b4.gz; ba.classes = classes;
            ba.cls_bias = pool_b;
            hipLaunchKernelGGL(fft4k_prep_kernel, dim3(F), dim3(kPrepWaves * 64), 0, (hipStream_t)stream, kernel, pool_w, F, K, gabor_bounds(K),
                               tab, Grow, f4.RG, (float2*)nullptr, ba);
            LEAF_LAUNCH_CHECK();
            return LEAF_OK;
        }
    }
    const BandLayout bl = band_layout(F, K, hop);
    if (!bl.stat) return LEAF_ERR_UNSUPPORTED;               // no band tasks for this geometry: every filter on full transforms
    const size_t need = leaf_fft_tables_bytes(F, K, hop);
    if (!workspace || need == 0 || workspace_bytes < need) return LEAF_ERR_WORKSPACE;
    if (misaligned(workspace) || misaligned(classes)) return LEAF_ERR_ALIGNMENT;
    const FftPlan fp = make_fft_plan(1, std::max(K, 2 * kFftN), F, K, hop);
    float* t = static_cast<float*>(workspace);
    float* Gz = t + align_up(fp.h_floats, 64);
    int* col_of = reinterpret_cast<int*>(Gz + align_up(fp.gz_floats, 64));
    float* bt = reinterpret_cast<float*>(col_of) + align_up((size_t)F, 64);
    BandTabArgs ba{};
    ba.hop = hop; ba.padL = fp.padL; ba.L = fp.L; ba.eps2 = kBandEps2; ba.eta = pool_b ? kBandEtaFree : kBandEta; ba.cross = pool_b ? 1 : 0;   // (pool_b = NULL: round 5's decision)
    ba.rec = reinterpret_cast<int*>(bt + bl.rec); ba.gz = bt + bl.gz; ba.classes = classes;
    ba.cls_bias = pool_b;
    hipLaunchKernelGGL(fft_prep_band_kernel, dim3(F, 2), dim3(kPrepWaves * 64), 0, (hipStream_t)stream, kernel, pool_w, F, K, fp.GZ,
                       gabor_bounds(K), reinterpret_cast<float2*>(t), Gz, col_of, ba);
    LEAF_LAUNCH_CHECK();
    return LEAF_OK;
}

int leaf_forward_prepared_f32(const float* x, int B, int T, const void* tables, size_t tables_bytes, const float* pool_b,
                              const float* alpha, const float* delta, const float* root, const float* ema_w, int F, int K,
                              int hop, int flags, float* out, void* workspace, size_t workspace_bytes, void* stream) {
    if (flags & LEAF_FLAG_PEAKNORM) return LEAF_ERR_UNSUPPORTED;         // no scale pre-pass on the prepared-tables path
    if (empty_batch(B, T, F, K, hop)) return LEAF_OK;
    if (!x || !tables || !pool_b || !out) return LEAF_ERR_NULL_POINTER;
    const bool use_pcen = (flags & LEAF_FLAG_PCEN) != 0;
    if (use_pcen && (!alpha || !delta || !root || !ema_w)) return LEAF_ERR_NULL_POINTER;
    int rc = check_shape(B, T, F, K, hop);
    if (rc != LEAF_OK) return rc;
    const FftPlan fp = make_fft_plan(B, T, F, K, hop);
    if (!fp.ok) return LEAF_ERR_BAD_ALGO;
    if (tables_bytes < leaf_fft_tables_bytes(F, K, hop)) return LEAF_ERR_WORKSPACE;
    const bool io_bf16 = (flags & LEAF_FLAG_IO_BF16) != 0;
    const uintptr_t io_mask = io_bf16 ? 1u : 3u;
    if ((reinterpret_cast<uintptr_t>(x) & io_mask) || (reinterpret_cast<uintptr_t>(out) & io_mask) || misaligned(workspace) ||
        misaligned(tables))
        return LEAF_ERR_ALIGNMENT;
    if (!workspace || workspace_bytes < (align_up(fp.part_floats, 64) + (LEAF_TRACE ? 16 * 64 * 2 : 0)) * 4) return LEAF_ERR_WORKSPACE;
    const int mode = (use_pcen ? 1 : 0) | ((flags & LEAF_FLAG_LOG1P) && !use_pcen ? 2 : 0) | (io_bf16 ? 4 : 0);
    // the edge tables of the band-limited filter tasks (they depend on the clip length) go behind the partial sums when the
    // workspace has the room (it has when sized by leaf_workspace_bytes as documented); otherwise full transforms
    const size_t part_end = align_up(fp.part_floats, 64) + (LEAF_TRACE ? 16 * 64 * 2 : 0);
    float* band_scratch = fp.band_dyn && workspace_bytes >= (part_end + fp.band_dyn) * 4 ? static_cast<float*>(workspace) + part_end : nullptr;
    return fft_forward(fp, x, io_bf16, B, T, nullptr, nullptr, pool_b, alpha, delta, root, ema_w, F, K, hop, mode, out,
                       static_cast<float*>(const_cast<void*>(tables)), static_cast<float*>(workspace), /*tables_ready=*/true,
                       (hipStream_t)stream, nullptr, nullptr,
                       auto_algo(B, T, F, K, hop, /*allow_small=*/false) == LEAF_ALGO_FFT_WG,   // the workgroup kernel from the batch AUTO takes it at
                       nullptr, band_scratch);
}

// ---- overlap-save backward: which geometries it covers, and its workspace layout (float offsets)
inline bool fft_backward_ok(const FftPlan& fp, int K, int hop) {
    return fp.ok && (K >= 224 || fft_static_geometry(K, hop));
}

struct FftBwdLayout {
    size_t R3, lone, Gz, col_of, part, raw, ema, gpre, rowsum, grow, dkpart, dwpart, dxblk, total;
    // band tasks of the static backward (leaf_band_bwd.hpp): records | G~ | G~2 | edge | edge2 | edge list; 0 floats where they do not apply
    size_t brec, bgz, bgz2, bedge, bedge2, belist;
};

// ---- workgroup-per-block backward (leaf_fft_wg_bwd.hpp): the static odd-window geometries; the only fused path that
// also yields dL/dx
struct FftWgBwdLaunch {
    FftKernel fn;
    int nw;
    size_t lds;
    bool block_dx;            // dL/dx from the workgroup-per-block kernel (G per block in LDS): grid over blocks, one plane per block
};
// blocks from which dL/dx comes from the workgroup-per-block kernels (G shared in LDS) rather than a block per wave: the
// thresholds of the parameter-gradient kernels
static bool wg_block_dx_enabled() {
    static const bool off = [] { const char* e = tools_env("LEAF_WG_BWD_DX"); return e && atoi(e) == 0; }();   // tools only: A/B
    return !off;
}
// blocks per CU (in sixteenths) from which the static 401 / 160 backward takes the workgroup kernel, whose narrow-band filters run as
// band tasks (leaf_band_bwd.hpp).  With dL/dx: always (profiles/r05/ab_band_dx.txt: 0.084 vs 0.087 ms at one clip, 0.085 vs 0.136 ms
// at 24); parameter gradients only: from 6/16 (0.063 ms flat from 1 to 24 clips against 0.041 / 0.060 / 0.104 / 0.124 ms of the
// per-wave kernel at 4 / 8 / 16 / 24 clips).  The thresholds assume that the default filters' classes hold (the decision itself is taken
// on the device, per call): a filterbank without narrow-band filters runs full tasks here below the 20/16 its own crossing was measured at.
#ifndef LEAF_WG_BWD_DX_BAND_SIXTEENTHS
#define LEAF_WG_BWD_DX_BAND_SIXTEENTHS 0       // blocks per CU (sixteenths) from which the static 401 / 160 backward WITH dL/dx takes the workgroup kernel (band tasks); 20: as without them
#endif
#ifndef LEAF_WG_BWD_BAND_SIXTEENTHS
#define LEAF_WG_BWD_BAND_SIXTEENTHS 6          // the same for the parameter gradients alone (below: the per-wave kernel); 20: as without band tasks
#endif
#ifndef LEAF_WG4K_BWD_BAND_SIXTEENTHS
#define LEAF_WG4K_BWD_BAND_SIXTEENTHS 8        // the same for the static 801 / 320 parameter-gradient backward on 4096-sample blocks (below: the 2048-sample kernels)
#endif
// the lowered thresholds hold only where band tasks CAN run in this call -- not with LEAF_FLAG_BWD_FULL_TRANSFORMS, more
// than kBandMaxFilters filters or clips whose edge frames band_edges() cannot table; leaf_backward_f32 says so here before it picks
// its kernels (the workspace layout does not depend on the pick), and such calls keep the 20/16 crossing measured without band tasks.
thread_local bool tl_bwd_band_possible = true;
inline int wg_bwd_sixteenths(int K, int hop, bool dx) {
    if (!band_geometry_ok(K, hop) || !LEAF_BAND_BWD || !tl_bwd_band_possible) return 20;
    return dx ? (LEAF_BAND_BWD_DX ? LEAF_WG_BWD_DX_BAND_SIXTEENTHS : 20) : LEAF_WG_BWD_BAND_SIXTEENTHS;
}
FftWgBwdLaunch pick_fft_wg_bwd_kernel(int K, int hop, bool dx, long long blocks = 0) {
    if (LEAF_FFT_FORCE_GENERIC) return {nullptr, 0, 0};
    if (dx) {
        if (!(fft_static_geometry(K, hop) && (K & 1))) return {nullptr, 0, 0};
        // (not K = 801: there the block-per-wave kernel measures 5 % faster, 1.89 vs 1.99 ms at 256 x 1 s)
        if (wg_block_dx_enabled() && K != 801 && blocks >= fft_wg_bwd_min_blocks(wg_bwd_sixteenths(K, hop, true)) &&
            fft_wg_bwd_dx_lds_bytes(12, K) <= (size_t)kMaxLds)
            return {as_fft_kernel(leaf_inst_fft_wg_bwd_dx(K)), 12, fft_wg_bwd_dx_lds_bytes(12, K), true};
        // below that: one wave per block, G in registers (leaf_fft_blk_bwd_dx_kernel)
        return {as_fft_kernel(leaf_inst_fft_blk_bwd_dx(K)), kBlkBwdWaves, fft_blk_bwd_lds_bytes(K)};
    }
    if (fft_static_geometry(K, hop) && (K & 1)) return {as_fft_kernel(leaf_inst_fft_wg_bwd(K)), 12, fft_wg_bwd_lds_bytes(12, K)};
    return {nullptr, 0, 0};
}
// any other window of the 2048-sample plan, odd or even: the run-time-geometry kernel (leaf_fft_wgg_bwd.hpp); parameter
// gradients only
FftWgBwdLaunch pick_fft_wgg_bwd_kernel(const FftPlan& fp, int K, int hop) {
    (void)hop;
    if (!fp.ok || K < 64 || K > 64 * 19) return {nullptr, 0, 0};
    static const int full_min = [] { const char* e = tools_env("LEAF_WGG_FULL"); return e ? atoi(e) : 11; }();   // as the forward's
    const int ni = fft_wgg_taps_per_lane(K);
    if (full_min > 0 && ni <= 10) {
        int nwf = 12;
        while (nwf > full_min && fft_wgg_lds_bytes_full(nwf, K) > (size_t)kMaxLds) --nwf;
        if (fft_wgg_lds_bytes_full(nwf, K) <= (size_t)kMaxLds)
            return {as_fft_kernel(leaf_inst_fft_wgg_bwd(ni, false)), nwf, fft_wgg_lds_bytes_full(nwf, K)};
    }
    int nw = 12;                                                          // the forward's row layout with the half-size scratch
    while (nw > 6 && fft_wgg_lds_bytes(nw, K) > (size_t)kMaxLds) --nw;
    if (fft_wgg_lds_bytes(nw, K) > (size_t)kMaxLds) return {nullptr, 0, 0};
    const FftWgLaunch fwd{nullptr, nw, fft_wgg_lds_bytes(nw, K)};
    return {as_fft_kernel(leaf_inst_fft_wgg_bwd(ni, true)), fwd.nw, fwd.lds};
}
static_assert(fft_wg_bwd_lds_bytes(12, 801) <= (size_t)kMaxLds && fft_blk_bwd_lds_bytes(801) <= (size_t)kMaxLds, "LDS budget");
// used for dL/dx always (nothing else fused yields it), and for the parameter gradients once every CU gets a block
bool fft_wg_bwd_use(const FftPlan& fp, int B, int K, int hop, bool need_dx) {
    return fp.ok && pick_fft_wg_bwd_kernel(K, hop, need_dx).fn != nullptr && (need_dx || (long long)B * fp.nblk >= fft_wg_bwd_min_blocks(wg_bwd_sixteenths(K, hop, false)));
}
// dL/dx for the other windows of the 2048-sample plan, odd or even, at every batch: the workgroup-per-block kernel with the
// block's G shared in LDS (leaf_fft_wgg_bwd_kernel<.., DX = true>), twelve-wave structure and dynamic filter queue.  (A
// block-per-wave run-time-geometry kernel served fewer than 10/16 block per CU until round 4; measured equal or slower there
// -- profiles/r04/dx_small_batches.txt -- and it carried 264..408 B of scratch per lane, so it is gone.)
FftWgBwdLaunch pick_fft_wgg_bwd_dx_kernel(const FftPlan& fp, int B, int K, int hop) {
    static const bool off = [] { const char* e = tools_env("LEAF_WGG_BWD_DX"); return e && atoi(e) == 0; }();   // tools only: A/B
    if (off || !fp.ok || K < 64 || K > 64 * 19 || pick_fft_wg_bwd_kernel(K, hop, true).fn) return {nullptr, 0, 0};
    (void)B;
    int nw = 12;
    while (nw > 6 && fft_wgg_bwd_dx_lds_bytes(nw, K) > (size_t)kMaxLds) --nw;
    const size_t lds = fft_wgg_bwd_dx_lds_bytes(nw, K);
    if (lds > (size_t)kMaxLds) return {nullptr, 0, 0};
    return {as_fft_kernel(leaf_inst_fft_wgg_bwd_dx(fft_wgg_taps_per_lane(K))), nw, lds};
}
// the run-time-geometry kernel: parameter gradients, once every CU gets a block
bool fft_wgg_bwd_use(const FftPlan& fp, int B, int K, int hop, bool need_dx) {
    static const bool off = [] { const char* e = tools_env("LEAF_WGG_BWD"); return e && atoi(e) == 0; }();   // tools only: A/B
    return !off && !need_dx && fp.ok && !pick_fft_wg_bwd_kernel(K, hop, false).fn && pick_fft_wgg_bwd_kernel(fp, K, hop).fn &&
           (long long)B * fp.nblk >= fft_wg_bwd_min_blocks(10);
}

#ifndef LEAF_BWD_ROWSUMS
#define LEAF_BWD_ROWSUMS 1             // 0: param_reduce_kernel re-reads the B x T' gradients for d pool_b (A/B)
#endif
FftBwdLayout fft_bwd_layout(const FftPlan& fp, int B, int F, bool need_dx) {
    FftBwdLayout L{};
    size_t o = 0;
    auto take = [&](size_t n) { const size_t at = o; o += align_up(n, 64); return at; };
    L.R3 = take((size_t)3 * F * kFftN);
    L.lone = take((size_t)3 * F * 2);                                     // even K: the unpaired tap and its mu / sigma derivatives
    L.Gz = take(fp.gz_floats);
    L.col_of = take((size_t)F);
    L.part = take(fp.part_floats);
    L.raw = take((size_t)B * F * fp.TP);
    L.ema = take((size_t)B * F * fp.TP);
    L.gpre = take((size_t)B * F * fp.TP);
    L.rowsum = take((size_t)B * F * 4);
    L.grow = take((size_t)B * F);                                         // sum_m g_pre per (clip, filter) row
    L.dkpart = take((size_t)B * fp.nblk * F * 2);
    L.dwpart = take((size_t)B * fp.nblk * F);
    L.dxblk = take(need_dx ? (size_t)B * fp.nblk * fp.nfq * kFftN : 0);    // per-(block, filter group) input gradients (the
                                                                        // workgroup-per-block kernels use one plane per block)
    if (fp.band_stat) {
        constexpr int K = 401;                                            // (band tasks exist for the 401 / 160 geometry only)
        L.brec = take((size_t)4 * F);
        L.bgz = take((size_t)F * band_gz_floats(K, 160));
        L.bgz2 = take((size_t)F * band_gz_floats(K, 160));
        L.bedge = take((size_t)F * 2 * kBandMaxEdge * 512);
        L.bedge2 = take((size_t)F * 2 * kBandMaxEdge * 512);
        L.belist = take((size_t)4 * kBandMaxEdge);
    }
    L.total = o;
    return L;
}

// ---- overlap-save backward on 4096-sample blocks (leaf_fft_wgg4k_bwd.hpp): odd windows 833..2049, parameter gradients,
// once every CU gets a block; run-time geometry
struct Fft4kBwdPlan {
    bool ok;
    bool stat;             // the static K = 801 / hop = 320 instance (register-gather pooling backward); false: run-time geometry
    bool dx;               // ... with dL/dx (static instance only): nine waves, the block's gradient spectra in LDS
    int L, nblk, TP, padL, RG, nw;
    size_t lds;
};
FftKernel pick_fft_wgg4k_bwd_kernel(int K) { return as_fft_kernel(leaf_inst_fft_wgg4k_bwd(fft_wgg4k_taps_per_lane(K))); }
Fft4kBwdPlan make_fft4k_bwd_plan(int B, int T, int F, int K, int hop, bool need_dx) {
    Fft4kBwdPlan bp{};
    static const bool off = [] { const char* e = tools_env("LEAF_4K_BWD"); return e && atoi(e) == 0; }();   // tools only: A/B
    static const int min_k = [] { const char* e = tools_env("LEAF_4K_BWD_MIN_K"); return e ? atoi(e) : 833; }();   // tools only
    // tools only: LEAF_4K_BWD_STATIC=0 keeps the 32 kHz geometry on the static 2048-sample kernel (A/B)
    static const bool stat_off = [] { const char* e = tools_env("LEAF_4K_BWD_STATIC"); return e && atoi(e) == 0; }();
    const bool stat = K == 801 && hop == 320 && !stat_off;
    // run-time geometry from K = 833; at K = 801 it measures slower than the static 2048-sample kernel (2.25 vs 2.09 ms)
    // dL/dx on 4096-sample blocks: the static instance only (tools: LEAF_4K_BWD_DX=0 keeps it on 2048-sample blocks, A/B)
    static const bool dx_off = [] { const char* e = tools_env("LEAF_4K_BWD_DX"); return e && atoi(e) == 0; }();
    if (off || fft4k_disabled() || (need_dx && (!stat || dx_off)) || !(K & 1) || (!stat && K < min_k) || K > 2049 || F > 65535)
        return bp;
    bp.stat = stat;
    bp.dx = need_dx;
    bp.padL = K / 2;
    bp.TP = (T - 1) / hop + 1;
    bp.L = stat ? 3200 : (kFft4N - K + 1) & ~1;                         // static: a multiple of the hop (the forward's plan)
    if ((bp.L + K - 2) / hop + 2 > 64) return bp;                        // g_pre of a block's frames: one per lane
    bp.nblk = ceil_div(T, bp.L);
    if ((long long)B * bp.nblk >= (1ll << 30) || (long long)B * bp.nblk < fft_wg_bwd_min_blocks(stat && !need_dx ? LEAF_WG4K_BWD_BAND_SIXTEENTHS : 8)) return bp;
    if (stat) {
        bp.RG = kWg4RowFloats;
        bp.nw = need_dx ? kWg4BwdDxWaves : LEAF_4K_BWD_NW;                           // half scratch + the two parity pooling rows per wave
        bp.lds = need_dx ? fft_wg4k_bwd_dx_lds_bytes(kWg4BwdDxWaves)
                         : (LEAF_4K_BWD_REGW && LEAF_4K_BWD_FULLSCR) ? fft_wg4k_lds_bytes(LEAF_4K_BWD_NW) : fft_wg4k_bwd_lds_bytes(LEAF_4K_BWD_NW);
        bp.ok = true;
        return bp;
    }
    bp.RG = fft_wgg4k_row_floats(K);
    bp.nw = 12;
    while (bp.nw > 6 && fft_wgg4k_lds_bytes(bp.nw, K, 0) > (size_t)kMaxLds) --bp.nw;     // no frame-sum array in the backward
    bp.lds = fft_wgg4k_lds_bytes(bp.nw, K, 0);
    bp.ok = bp.lds <= (size_t)kMaxLds;
    return bp;
}
struct Fft4kBwdLayout {
    size_t tab3, grow, part, raw, ema, gpre, rowsum, gsrow, dkpart, dwpart, col_of, dxblk, total;
    size_t brec, bgz, bgz2, bedge, bedge2, belist;   // band tasks of the static 32 kHz backward (leaf_band_bwd.hpp); 0 where they do not apply
};
Fft4kBwdLayout fft4k_bwd_layout(const Fft4kBwdPlan& bp, int B, int F) {
    Fft4kBwdLayout L{};
    size_t o = 0;
    auto take = [&](size_t n) { const size_t at = o; o += align_up(n, 64); return at; };
    L.tab3 = take((size_t)3 * F * kFft4TabFloats);
    L.grow = take((size_t)F * 2 * bp.RG + kFft4WtFloats);                // + the shared twiddle table of the static forward kernel
    L.part = take((size_t)B * bp.TP * 2 * F);
    L.raw = take((size_t)B * F * bp.TP);
    L.ema = take((size_t)B * F * bp.TP);
    L.gpre = take((size_t)B * F * bp.TP);
    L.rowsum = take((size_t)B * F * 4);
    L.gsrow = take((size_t)B * F);                                        // sum_m g_pre per (clip, filter) row
    L.dkpart = take((size_t)B * bp.nblk * F * 2);
    L.dwpart = take((size_t)B * bp.nblk * F);
    L.col_of = take((size_t)F);
    L.dxblk = take(bp.dx ? (size_t)B * bp.nblk * kFft4N : 0);            // per-block input gradients, 4096 samples each
    if (bp.stat && F <= kBandMaxFilters) {                              // (with dL/dx unused: the two layouts differ by dxblk only)
        L.brec = take((size_t)4 * F);
        L.bgz = take((size_t)F * band4k_gz_floats(801, 320));
        L.bgz2 = take((size_t)F * band4k_gz_floats(801, 320));
        L.bedge = take((size_t)F * kBandMaxEdge * 512);
        L.bedge2 = take((size_t)F * kBandMaxEdge * 512);
        L.belist = take((size_t)4 * kBandMaxEdge);
    }
    L.total = o;
    return L;
}

// Which backward implementation serves a call (the same decision sizes the workspace and dispatches the kernels).
enum BwdPath { BWD_PATH_FFT = 0, BWD_PATH_MFMA = 1, BWD_PATH_STAGED = 2, BWD_PATH_FFT4K = 3 };
static BwdPath bwd_path(int B, int T, int F, int K, int hop, int flags, bool need_dx) {
    if (!(flags & (LEAF_FLAG_BWD_STAGED | LEAF_FLAG_BWD_MFMA))) {
        if (make_fft4k_bwd_plan(B, T, F, K, hop, need_dx).ok) return BWD_PATH_FFT4K;
        const FftPlan fp = make_fft_plan(B, T, F, K, hop);
        if (fft_backward_ok(fp, K, hop) &&
            (!need_dx || fft_wg_bwd_use(fp, B, K, hop, true) || pick_fft_wgg_bwd_dx_kernel(fp, B, K, hop).fn))
            return BWD_PATH_FFT;
    }
    if (!need_dx && !(flags & LEAF_FLAG_BWD_STAGED)) {
        const FusedPlan pl = make_plan(B, T, F, K, hop);
        if (make_bwd_plan(pl, T).ok) return BWD_PATH_MFMA;
    }
    return BWD_PATH_STAGED;
}

size_t leaf_backward_workspace_bytes(int B, int T, int F, int K, int hop, int flags, int need_dx) {
    if (check_shape(B, T, F, K, hop) != LEAF_OK) return 0;
    switch (bwd_path(B, T, F, K, hop, flags, need_dx != 0)) {
        case BWD_PATH_FFT: return fft_bwd_layout(make_fft_plan(B, T, F, K, hop), B, F, need_dx != 0).total * 4;
        case BWD_PATH_FFT4K: return fft4k_bwd_layout(make_fft4k_bwd_plan(B, T, F, K, hop, need_dx != 0), B, F).total * 4;
        case BWD_PATH_MFMA: {
            const FusedPlan pl = make_plan(B, T, F, K, hop);
            return bwd_layout(pl, make_bwd_plan(pl, T), B, T, F, num_cus()).total * 4;
        }
        default: break;
    }
    const int TP = (T - 1) / hop + 1;
    const size_t fl = align_up((size_t)2 * F * K, 64) * 2 /* taps, dtaps unused slot */ + align_up((size_t)F * K, 64) * 2 +
                      align_up((size_t)B * 2 * F * T, 64) + align_up((size_t)B * F * T, 64) +
                      align_up((size_t)B * F * TP, 64) * 3 + align_up((size_t)B * F * 4, 64) +
                      align_up((size_t)B * 2 * F * K, 64);
    return fl * 4;
}

int leaf_backward_f32(const float* x, int B, int T, const float* kernel, const float* pool_w, const float* pool_b,
                      const float* alpha, const float* delta, const float* root, const float* ema_w, int F, int K, int hop,
                      int flags, const float* grad_out, const float* pooled_raw, float* g_kernel, float* g_pool_w,
                      float* g_pool_b, float* g_alpha, float* g_delta, float* g_root, float* g_ema_w, float* g_x,
                      void* workspace, size_t workspace_bytes, void* stream) {
    if (empty_batch(B, T, F, K, hop)) {
        // the sum over zero clips: every parameter gradient is exactly zero (what autograd returns for the reference)
        if (!g_kernel || !g_pool_w || !g_pool_b) return LEAF_ERR_NULL_POINTER;
        const bool pc = (flags & LEAF_FLAG_PCEN) != 0;
        if (pc && (!g_alpha || !g_delta || !g_root || !g_ema_w)) return LEAF_ERR_NULL_POINTER;
        hipStream_t s0 = (hipStream_t)stream;
        bool ok = hipMemsetAsync(g_kernel, 0, (size_t)2 * F * 4, s0) == hipSuccess;
        ok = ok && hipMemsetAsync(g_pool_w, 0, (size_t)F * 4, s0) == hipSuccess;
        ok = ok && hipMemsetAsync(g_pool_b, 0, (size_t)F * 4, s0) == hipSuccess;
        if (pc)
            for (float* g : {g_alpha, g_delta, g_root, g_ema_w}) ok = ok && hipMemsetAsync(g, 0, (size_t)F * 4, s0) == hipSuccess;
        return ok ? LEAF_OK : LEAF_ERR_LAUNCH;
    }
    if (!x || !kernel || !pool_w || !pool_b || !grad_out || !g_kernel || !g_pool_w || !g_pool_b) return LEAF_ERR_NULL_POINTER;
    const bool use_pcen = (flags & LEAF_FLAG_PCEN) != 0;
    if (use_pcen && (!alpha || !delta || !root || !ema_w || !g_alpha || !g_delta || !g_root || !g_ema_w))
        return LEAF_ERR_NULL_POINTER;
    int rc = check_shape(B, T, F, K, hop);
    if (rc != LEAF_OK) return rc;
    if (2 * F > 65535 || B > 65535) return LEAF_ERR_BAD_SHAPE;
    const BwdPath path = bwd_path(B, T, F, K, hop, flags, g_x != nullptr);
    if (!workspace || workspace_bytes < leaf_backward_workspace_bytes(B, T, F, K, hop, flags, g_x != nullptr))
        return LEAF_ERR_WORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    const int TP = (T - 1) / hop + 1;
    const int padL = K / 2 + K % 2 - 1;
    const int mode = use_pcen ? 1 : 0;
    float* ws = static_cast<float*>(workspace);
    if (path == BWD_PATH_FFT4K) {
        // ---- overlap-save backward on 4096-sample blocks: long odd windows, parameter gradients
        const Fft4kBwdPlan bp = make_fft4k_bwd_plan(B, T, F, K, hop, g_x != nullptr);
        const Fft4kBwdLayout L = fft4k_bwd_layout(bp, B, F);
        float* tab3 = ws + L.tab3; float* Grow = ws + L.grow; float* part = ws + L.part; float* raw = ws + L.raw;
        float* ema = ws + L.ema; float* gpre = ws + L.gpre; float* rowsum = ws + L.rowsum; float* dkpart = ws + L.dkpart;
        float* dwpart = ws + L.dwpart; int* col_of = reinterpret_cast<int*>(ws + L.col_of);
        // 1. tables: 4096-point real spectra of w, dw/dmu, dw/dsigma (+ the D tables of w) and the de-interleaved pooling rows
        float2* Wt = reinterpret_cast<float2*>(Grow + (size_t)F * 2 * bp.RG);
        // band tasks of the static 32 kHz backward (leaf_band_bwd.hpp): decision by fft4k_prep_kernel's workgroups (f, 0), G~, G~2 and
        // the edge tables by fft4k_band_tab_kernel, from the parameters of this call
        static const bool band_bwd_off4 = [] { const char* e = tools_env("LEAF_BAND_BWD"); return e && atoi(e) == 0; }();   // tools only: A/B
        BandParams band{};
        BandTabArgs ba{};
        const bool band_bwd = LEAF_BAND_BWD && !band_bwd_off4 && !(flags & LEAF_FLAG_BWD_FULL_TRANSFORMS) && bp.stat && !bp.dx && L.bgz2 &&
                              LEAF_4K_BWD_REGW && LEAF_4K_BWD_FULLSCR && bp.lds + band_lds_bytes(F) <= (size_t)kMaxLds &&
                              band_edges(T, K, hop, bp.L, bp.padL, band, ba.e);
        if (band_bwd) {
            ba.T = T; ba.L = bp.L; ba.hop = hop; ba.padL = bp.padL; ba.eps2 = kBandEps2; ba.eta = (flags & LEAF_FLAG_BWD_STRICT_BAND_CLASSES) ? kBandEta : kBandEtaFree;
            ba.bwd_slabs = 2;                     // (a backward launch: the class decision also asks band_deriv_fits, leaf_band.hpp; no extra grid rows in this kernel)
            ba.rec = reinterpret_cast<int*>(ws + L.brec); ba.gz = ws + L.bgz; ba.gz2 = ws + L.bgz2; ba.edge = ws + L.bedge;
            ba.edge2 = ws + L.bedge2; ba.elist = reinterpret_cast<int*>(ws + L.belist); ba.n_edge = band.n_edge;
            band.rec = ba.rec; band.gz = ba.gz; band.gz2 = ba.gz2; band.edge = ba.edge; band.edge2 = ba.edge2; band.elist = ba.elist;
            if (!(flags & LEAF_FLAG_BWD_STRICT_BAND_CLASSES)) { band.bias = pool_b; band.smax = 2.0f; }   // the forward's bias-aware class decision (leaf_band.hpp)
        }
        hipLaunchKernelGGL(fft4k_prep_kernel, dim3(F, 3), dim3(kPrepWaves * 64), 0, st, kernel, pool_w, F, K, gabor_bounds(K), tab3,
                           Grow, bp.RG, Wt, ba);
        LEAF_LAUNCH_CHECK();
        if (band_bwd) hipLaunchKernelGGL(fft4k_band_tab_kernel, dim3(F, 1 + band.n_edge), dim3(kPrepWaves * 64), 0, st, pool_w, F, K, ba);
        LEAF_LAUNCH_CHECK();
        hipLaunchKernelGGL(iota_kernel, dim3(ceil_div(F, 256)), dim3(256), 0, st, col_of, F);
        LEAF_LAUNCH_CHECK();
        FftParams q{};
        q.x = x; q.io_bf16 = 0; q.H = reinterpret_cast<const float2*>(tab3); q.Gz = Grow; q.part = part;
        q.B = B; q.T = T; q.TP = bp.TP; q.F = F; q.K = K; q.hop = hop; q.padL = bp.padL; q.L = bp.L; q.nblk = bp.nblk;
        q.nslot = 2; q.GZ = bp.RG;
        q.lone = reinterpret_cast<const float*>(Wt);                      // (read by the static forward kernel when it recomputes below)
        const dim3 grid(std::max(1, std::min(B * bp.nblk, num_cus())));
        const float* raw_in = pooled_raw;              // saved by leaf_forward_save_f32, else recomputed here
        if (!raw_in) {
            // the forward with its own wave count and LDS (run-time geometry: it parks frame sums between the halves)
            const int fbn = bp.stat ? 0 : fft_wgg4k_frame_floats(K, hop);
            int fnw = bp.stat ? LEAF_4K_FWD_NW : 12;
            while (!bp.stat && fnw > 6 && fft_wgg4k_lds_bytes(fnw, K, fbn) > (size_t)kMaxLds) --fnw;
            const size_t flds = bp.stat ? fft_wg4k_lds_bytes(LEAF_4K_FWD_NW) : fft_wgg4k_lds_bytes(fnw, K, fbn);
            if (flds > (size_t)kMaxLds) return LEAF_ERR_BAD_ALGO;
            FftKernel kf = bp.stat ? as_fft_kernel(leaf_inst_fft_wg4k()) : pick_fft_wgg4k_kernel(K);
            q.NT = fbn;
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kf), hipFuncAttributeMaxDynamicSharedMemorySize, (int)flds);
            hipLaunchKernelGGL(kf, grid, dim3(fnw * 64), flds, st, q);
            LEAF_LAUNCH_CHECK();
            q.NT = 0;
            launch_fft_finalize(FinParams{part, F, TP, SlotGeom{bp.L, bp.padL, K, hop, T, 2}, pool_b, alpha, delta, root, ema_w, 1e-12f, 8,
                                          raw, raw, nullptr}, B, OwnedClips{}, st);
            LEAF_LAUNCH_CHECK();
            raw_in = raw;
        }
        // 2. floor + PCEN backward per (b,f) row
        hipLaunchKernelGGL(pcen_bwd_scan_kernel, dim3(ceil_div(B * F, 4)), dim3(256), 0, st, raw_in, grad_out, B * F, F, TP, alpha,
                           delta, root, ema_w, 1e-12f, mode, ema, gpre, rowsum, (const int*)nullptr, 0, (float*)nullptr, LEAF_BWD_ROWSUMS ? ws + L.gsrow : nullptr);
        LEAF_LAUNCH_CHECK();
        // 3. per-(block, filter) partial gradients
        q.gpre = gpre; q.pool_w = pool_w; q.dkpart = dkpart; q.dwpart = dwpart; q.part = nullptr;
        FftKernel kb = bp.dx ? as_fft_kernel(leaf_inst_fft_wg4k_bwd_dx())
                             : bp.stat ? as_fft_kernel(leaf_inst_fft_wg4k_bwd()) : pick_fft_wgg4k_bwd_kernel(K);
        if (bp.dx) q.part = ws + L.dxblk;                                 // [block][4096] input gradients, un-rotated
        size_t blds = bp.lds;
        if (band_bwd) {
            band.lds_off = (int)(blds / 4);
            blds += band_lds_bytes(F);
            q.band = band;
        }
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kb), hipFuncAttributeMaxDynamicSharedMemorySize, (int)blds);
        hipLaunchKernelGGL(kb, grid, dim3(bp.nw * 64), blds, st, q);
        LEAF_LAUNCH_CHECK();
        if (bp.dx) {
            hipLaunchKernelGGL(fft_dx_gather_kernel, dim3(ceil_div(T, 1024), B), dim3(256), 0, st, ws + L.dxblk, T, bp.nblk, 1, bp.L,
                               bp.padL, g_x, kFft4N);
            LEAF_LAUNCH_CHECK();
        }
        // 4. reductions over blocks and the batch, clamp sub-gradients
        // (the per-block (d mu, d sigma) partials are summed by param_reduce_kernel below: one launch less)
        hipLaunchKernelGGL(param_reduce_kernel, dim3(F), dim3(kParamRedThreads), 0, st, gpre, (const float*)nullptr,
                           (const float*)nullptr, rowsum, pool_w, B, F, TP, K, mode, dwpart, B * bp.nblk, F, col_of, g_pool_w,
                           g_pool_b, g_alpha, g_delta, g_root, g_ema_w,
                           dkpart, B * bp.nblk, kernel, gabor_bounds(K), g_kernel, LEAF_BWD_ROWSUMS ? ws + L.gsrow : nullptr);
        LEAF_LAUNCH_CHECK();
        return LEAF_OK;
    }
    {
        // ---- overlap-save backward: odd windows the FFT forward is chosen for (K >= 224), dL/dx not requested
        const FftPlan fp = make_fft_plan(B, T, F, K, hop);
        struct BandPossible {                                   // scoped: the thresholds of THIS call (wg_bwd_sixteenths)
            bool prev;
            explicit BandPossible(bool v) : prev(tl_bwd_band_possible) { tl_bwd_band_possible = v; }
            ~BandPossible() { tl_bwd_band_possible = prev; }
        };
        BandParams band_probe{};
        BandEdge edge_probe[kBandMaxEdge];
        const BandPossible band_possible(fp.ok && !(flags & LEAF_FLAG_BWD_FULL_TRANSFORMS) && (K & 1) && F <= kBandMaxFilters &&
                                         fp.nslot == 2 && band_edges(T, K, hop, fp.L, fp.padL, band_probe, edge_probe));
        if (path == BWD_PATH_FFT) {
            const FftBwdLayout L = fft_bwd_layout(fp, B, F, g_x != nullptr);
            float* R3 = ws + L.R3; float* Gz = ws + L.Gz; int* col_of = reinterpret_cast<int*>(ws + L.col_of);
            float* part = ws + L.part; float* raw = ws + L.raw; float* ema = ws + L.ema; float* gpre = ws + L.gpre;
            float* rowsum = ws + L.rowsum; float* dkpart = ws + L.dkpart; float* dwpart = ws + L.dwpart;
            // 1. tables: real spectra of w, dw/dmu, dw/dsigma and the pooling rows -- with the band tasks of the static backward
            // (leaf_band_bwd.hpp: the filters the forward runs on 256- / 512-point transforms get their parameter gradients at the
            // decimated rate too) one launch of fft_prep_band_kernel builds them together with the decision, G~, G~2 and the edge
            // tables, from the parameters of THIS call (the decision the forward took from the same parameters)
            static const bool band_bwd_off = [] { const char* e = tools_env("LEAF_BAND_BWD"); return e && atoi(e) == 0; }();   // tools only: A/B
            BandParams band{};
            BandTabArgs ba{};
            // (with dL/dx: in the workgroup-per-block kernel, whose band tasks add their members' shares to the block's G)
            const FftWgBwdLaunch bwl = pick_fft_wg_bwd_kernel(K, hop, g_x != nullptr, (long long)B * fp.nblk);
            const bool band_bwd = LEAF_BAND_BWD && !band_bwd_off && !(flags & LEAF_FLAG_BWD_FULL_TRANSFORMS) && L.bgz2 && (K & 1) &&
                                  F <= kBandMaxFilters && fp.nslot == 2 && fft_wg_bwd_use(fp, B, K, hop, g_x != nullptr) &&
                                  (g_x ? LEAF_BAND_BWD_DX && bwl.block_dx : true) &&
                                  bwl.lds + band_lds_bytes(F) <= (size_t)kMaxLds && band_edges(T, K, hop, fp.L, fp.padL, band, ba.e);
            if (band_bwd) {
                ba.T = T; ba.L = fp.L; ba.hop = hop; ba.padL = fp.padL; ba.eps2 = kBandEps2; ba.eta = (flags & LEAF_FLAG_BWD_STRICT_BAND_CLASSES) ? kBandEta : kBandEtaFree;
                ba.rec = reinterpret_cast<int*>(ws + L.brec); ba.gz = ws + L.bgz; ba.gz2 = ws + L.bgz2; ba.edge = ws + L.bedge;
                ba.edge2 = ws + L.bedge2; ba.elist = reinterpret_cast<int*>(ws + L.belist); ba.n_edge = band.n_edge;
                ba.bwd_slabs = 1;
                hipLaunchKernelGGL(fft_prep_band_kernel, dim3(F, 2 + band.n_edge + 2), dim3(kPrepWaves * 64), 0, st, kernel, pool_w, F, K, fp.GZ,
                                   gabor_bounds(K), reinterpret_cast<float2*>(R3), Gz, col_of, ba);
                band.rec = ba.rec; band.gz = ba.gz; band.gz2 = ba.gz2; band.edge = ba.edge; band.edge2 = ba.edge2; band.elist = ba.elist;
                if (!(flags & LEAF_FLAG_BWD_STRICT_BAND_CLASSES)) { band.bias = pool_b; band.smax = 2.0f; }   // the forward's bias-aware class decision
            } else {
                hipLaunchKernelGGL(fft_prep_kernel, dim3(F, 3), dim3(kPrepWaves * 64), 0, st, kernel, pool_w, F, K, fp.GZ,
                                   gabor_bounds(K), 1, reinterpret_cast<float2*>(R3), Gz, col_of, (K & 1) ? (float*)nullptr : ws + L.lone);
            }
            LEAF_LAUNCH_CHECK();
            FftParams q{};
            q.x = x; q.io_bf16 = 0; q.H = reinterpret_cast<const float2*>(R3); q.Gz = Gz; q.part = part;
            q.lone = (K & 1) ? nullptr : ws + L.lone;
            q.B = B; q.T = T; q.TP = fp.TP; q.F = F; q.K = K; q.hop = hop; q.padL = fp.padL;
            q.L = fp.L; q.nblk = fp.nblk; q.GZ = fp.GZ; q.nslot = fp.nslot; q.g_bufs = fp.g_bufs; q.NT = fp.NT; q.fq = fp.fq; q.nfq = fp.nfq;
            q.scr_floats = fp.scr_floats; q.total_tasks = B * fp.nblk * fp.nfq;
            q.rot = K / 2;
            const dim3 grid(std::max(1, std::min(ceil_div(q.total_tasks, kFftWaves), num_cus())));
            const float* raw_in = pooled_raw;          // saved by leaf_forward_save_f32, else recomputed here
            if (!raw_in) {
                FftKernel kf = pick_fft_kernel(fp, K, hop, false);
                (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kf), hipFuncAttributeMaxDynamicSharedMemorySize, (int)fp.lds);
                hipLaunchKernelGGL(kf, grid, dim3(kFftWaves * 64), fp.lds, st, q);
                LEAF_LAUNCH_CHECK();
                launch_fft_finalize(FinParams{part, F, TP, SlotGeom{fp.L, fp.padL, K, hop, T, fp.nslot}, pool_b, alpha, delta, root, ema_w,
                                              1e-12f, 8, raw, raw, nullptr}, B, OwnedClips{}, st);
                LEAF_LAUNCH_CHECK();
                raw_in = raw;
            }
            // 2. floor + PCEN backward per (b,f) row
            hipLaunchKernelGGL(pcen_bwd_scan_kernel, dim3(ceil_div(B * F, 4)), dim3(256), 0, st, raw_in, grad_out, B * F, F, TP,
                               alpha, delta, root, ema_w, 1e-12f, mode, ema, gpre, rowsum, (const int*)nullptr, 0,
                               (float*)nullptr, LEAF_BWD_ROWSUMS ? ws + L.grow : nullptr);
            LEAF_LAUNCH_CHECK();
            // 3. filterbank recompute + transposed pooling + second transform: per-block (d mu, d sigma, d pool_w)
            q.gpre = gpre; q.pool_w = pool_w; q.dkpart = dkpart; q.dwpart = dwpart; q.part = nullptr;
            if (fft_wg_bwd_use(fp, B, K, hop, g_x != nullptr)) {
                // workgroup-per-block backward; with g_x the per-block input gradients go to dxblk and are gathered below
                FftWgBwdLaunch wl = pick_fft_wg_bwd_kernel(K, hop, g_x != nullptr, (long long)B * fp.nblk);
                q.part = g_x ? ws + L.dxblk : nullptr;
                if (band_bwd) {
                    band.lds_off = (int)(wl.lds / 4);
                    wl.lds += band_lds_bytes(F);
                    q.band = band;
                }
                (void)hipFuncSetAttribute(reinterpret_cast<const void*>(wl.fn), hipFuncAttributeMaxDynamicSharedMemorySize, (int)wl.lds);
                // dx kernels: one (block, group) per WAVE, or (block_dx) a workgroup per block like the parameter-gradient kernel
                const int wgs = g_x && !wl.block_dx ? ceil_div(q.total_tasks, wl.nw) : B * fp.nblk;
                hipLaunchKernelGGL(wl.fn, dim3(std::max(1, std::min(wgs, num_cus()))), dim3(wl.nw * 64), wl.lds, st, q);
                LEAF_LAUNCH_CHECK();
                if (g_x) {
                    hipLaunchKernelGGL(fft_dx_gather_kernel, dim3(ceil_div(T, 1024), B), dim3(256), 0, st, ws + L.dxblk, T, fp.nblk,
                                       wl.block_dx ? 1 : fp.nfq, fp.L, fp.padL, g_x);
                    LEAF_LAUNCH_CHECK();
                }
            } else if (g_x && pick_fft_wgg_bwd_dx_kernel(fp, B, K, hop).fn) {
                // dL/dx on a window without a static instance: workgroup per block, G in LDS
                const FftWgBwdLaunch wl = pick_fft_wgg_bwd_dx_kernel(fp, B, K, hop);
                q.part = ws + L.dxblk;                                    // [block][2048]: one plane per block
                (void)hipFuncSetAttribute(reinterpret_cast<const void*>(wl.fn), hipFuncAttributeMaxDynamicSharedMemorySize, (int)wl.lds);
                hipLaunchKernelGGL(wl.fn, dim3(std::max(1, std::min(B * fp.nblk, num_cus()))), dim3(wl.nw * 64), wl.lds, st, q);
                LEAF_LAUNCH_CHECK();
                hipLaunchKernelGGL(fft_dx_gather_kernel, dim3(ceil_div(T, 1024), B), dim3(256), 0, st, ws + L.dxblk, T, fp.nblk, 1, fp.L,
                                   fp.padL, g_x);
                LEAF_LAUNCH_CHECK();
            } else if (g_x) {
                return LEAF_ERR_BAD_ALGO;                                 // bwd_path() admits the FFT path only with a dL/dx kernel
            } else if (fft_wgg_bwd_use(fp, B, K, hop, g_x != nullptr)) {
                const FftWgBwdLaunch wl = pick_fft_wgg_bwd_kernel(fp, K, hop);
                (void)hipFuncSetAttribute(reinterpret_cast<const void*>(wl.fn), hipFuncAttributeMaxDynamicSharedMemorySize, (int)wl.lds);
                hipLaunchKernelGGL(wl.fn, dim3(std::max(1, std::min(B * fp.nblk, num_cus()))), dim3(wl.nw * 64), wl.lds, st, q);
                LEAF_LAUNCH_CHECK();
            } else {
                FftKernel kb = pick_fft_kernel(fp, K, hop, true);
                (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kb), hipFuncAttributeMaxDynamicSharedMemorySize, (int)fp.lds);
                hipLaunchKernelGGL(kb, grid, dim3(kFftWaves * 64), fp.lds, st, q);
                LEAF_LAUNCH_CHECK();
            }
            // 4. reductions over blocks and the batch, clamp sub-gradients
            // (the per-block (d mu, d sigma) partials are summed by param_reduce_kernel below: one launch less)
            hipLaunchKernelGGL(param_reduce_kernel, dim3(F), dim3(kParamRedThreads), 0, st, gpre, (const float*)nullptr,
                               (const float*)nullptr, rowsum, pool_w, B, F, TP, K, mode, dwpart, B * fp.nblk, F, col_of,
                               g_pool_w, g_pool_b, g_alpha, g_delta, g_root, g_ema_w,
                               dkpart, B * fp.nblk, kernel, gabor_bounds(K), g_kernel, LEAF_BWD_ROWSUMS ? ws + L.grow : nullptr);
            LEAF_LAUNCH_CHECK();
            return LEAF_OK;
        }
    }
    {
        // ---- fused backward (MFMA): used whenever the geometry fits and dL/dx is not requested
        const FusedPlan pl = make_plan(B, T, F, K, hop);
        const BwdPlan bp = make_bwd_plan(pl, T);
        if (path == BWD_PATH_MFMA) {
            const int cus = num_cus();
            const BwdLayout L = bwd_layout(pl, bp, B, T, F, cus);
            float* W = ws + L.W; float* G = ws + L.G; float* Gs = ws + L.Gs;
            int* meta = reinterpret_cast<int*>(ws + L.meta);
            int* perm = meta; int* col_of = meta + pl.FP; int* tile_ks = meta + 2 * pl.FP;
            float* part = ws + L.part; float* raw = ws + L.raw; float* ema = ws + L.ema; float* gpre = ws + L.gpre;
            float* gcols = ws + L.gcols; float* rowsum = ws + L.rowsum; float* dwpart = ws + L.dwpart;
            float* dHpart = ws + L.dHpart; float* dY = ws + L.dY;
            const int Rp = 16 * bp.NKT;
            // 1. tables, forward recompute up to the pre-floor pooled value
            hipLaunchKernelGGL(fused_prep_kernel, dim3(ceil_div(pl.R * 2 * pl.FP + pl.FP * pl.GJ, 256)), dim3(256), 0, st,
                               kernel, pool_w, F, pl.FP, K, pl.R, pl.GJ, gabor_bounds(K), W, G, Gs, perm, col_of, tile_ks);
            LEAF_LAUNCH_CHECK();
            FusedParams prm = base_fused_params(pl, B, T, F, K, hop, -1);
            prm.x = x; prm.W = W; prm.G = G; prm.tile_ks = tile_ks; prm.part = part;
            const float* raw_in = pooled_raw;          // saved by leaf_forward_save_f32, else recomputed here
            if (!raw_in) {
                if (launch_fused_groups(prm, pl, st) != hipSuccess) return LEAF_ERR_LAUNCH;
                hipLaunchKernelGGL(finalize_kernel, dim3(B, ceil_div(F, kFinGroup)), dim3(kFinGroup * 64), (size_t)kFinGroup * (kFinStride + 3) * 4, st, part, F, pl.FP, TP,
                                   pl.noff, pl.q_lo, pl.q_hi, SlotGeom{}, col_of, pool_b, alpha, delta, root, ema_w, 1e-12f, 8, raw,
                                   (float*)nullptr);
                LEAF_LAUNCH_CHECK();
                raw_in = raw;
            }
            // 2. floor + PCEN backward per (b,f) row
            if (hipMemsetAsync(gcols, 0, (size_t)B * TP * pl.FP * 4, st) != hipSuccess) return LEAF_ERR_LAUNCH;
            hipLaunchKernelGGL(pcen_bwd_scan_kernel, dim3(ceil_div(B * F, 4)), dim3(256), 0, st, raw_in, grad_out, B * F, F, TP,
                               alpha, delta, root, ema_w, 1e-12f, mode, ema, gpre, rowsum, col_of, pl.FP, gcols);
            LEAF_LAUNCH_CHECK();
            // 3. filterbank recompute with the backward epilogue: dY (time-major) and d pool_w partials
            if (hipMemsetAsync(dwpart, 0, (size_t)cus * kWavesPerWG * pl.FP * 4, st) != hipSuccess) return LEAF_ERR_LAUNCH;
            if (hipMemsetAsync(dHpart, 0, (size_t)cus * Rp * 2 * pl.FP * 4, st) != hipSuccess) return LEAF_ERR_LAUNCH;
            prm.Gs = Gs; prm.gcols = gcols; prm.dY = dY; prm.dwpart = dwpart; prm.part = nullptr;
            if (launch_fused_groups(prm, pl, st) != hipSuccess) return LEAF_ERR_LAUNCH;
            // 4. tap gradients: dH = S^T dY on the MFMA, then chain to (mu, sigma)
            DtapsParams dp{};
            dp.x = x; dp.dY = dY; dp.tile_ks = tile_ks; dp.dHpart = dHpart;
            dp.B = B; dp.T = T; dp.FP = pl.FP; dp.K = K; dp.Hf = pl.Hf; dp.xshift = pl.xshift;
            dp.NKT = bp.NKT; dp.NW = bp.NW; dp.NS = bp.NS; dp.HPc = bp.HPc; dp.XSC = bp.XSC; dp.LD = bp.LD;
            dp.nch = bp.nch; dp.total_chunks = B * bp.nch;
            if (pl.groups_main > 0) {
                dp.tile_base = 0;
                const int gx = std::max(1, std::min(dp.total_chunks, std::max(1, cus / pl.groups_main)));
                if (launch_dtaps(dp, pl.rt_main, bp.TPW, pl.groups_main, bp.lds, gx, st) != hipSuccess) return LEAF_ERR_LAUNCH;
            }
            if (pl.rt_rem > 0) {
                dp.tile_base = pl.groups_main * pl.rt_main;
                const int gx = std::max(1, std::min(dp.total_chunks, cus));
                if (launch_dtaps(dp, pl.rt_rem, bp.TPW, 1, bp.lds, gx, st) != hipSuccess) return LEAF_ERR_LAUNCH;
            }
            {
                // slab 0 doubles as the reduction target: reduce slabs 1.. into a scratch slab placed after the last one
                const size_t slab = (size_t)Rp * 2 * pl.FP;
                float* dHsum = dHpart + (size_t)cus * slab;
                hipLaunchKernelGGL(dh_reduce_kernel, dim3((unsigned)((slab + 255) / 256)), dim3(256), 0, st, dHpart, cus, slab,
                                   dHsum);
                LEAF_LAUNCH_CHECK();
                hipLaunchKernelGGL(dkernel_fused_kernel, dim3(F), dim3(256), 0, st, dHsum, 1, Rp, W, pl.R, pl.FP, col_of,
                                   kernel, F, gabor_bounds(K), g_kernel);
            }
            LEAF_LAUNCH_CHECK();
            // 5. parameter sums over the batch
            hipLaunchKernelGGL(param_reduce_kernel, dim3(F), dim3(kParamRedThreads), 0, st, gpre, (const float*)nullptr,
                               (const float*)nullptr, rowsum, pool_w, B, F, TP, K, mode, dwpart, cus * kWavesPerWG, pl.FP,
                               col_of, g_pool_w, g_pool_b, g_alpha, g_delta, g_root, g_ema_w);
            LEAF_LAUNCH_CHECK();
            return LEAF_OK;
        }
    }
    float* taps = ws;                         ws += align_up((size_t)2 * F * K, 64) * 2;
    float* g = ws;                            ws += align_up((size_t)F * K, 64);
    float* dg = ws;                           ws += align_up((size_t)F * K, 64);
    float* y = ws;                            ws += align_up((size_t)B * 2 * F * T, 64);
    float* e = ws;                            ws += align_up((size_t)B * F * T, 64);
    float* raw = ws;                          ws += align_up((size_t)B * F * TP, 64);
    float* ema = ws;                          ws += align_up((size_t)B * F * TP, 64);
    float* gpre = ws;                         ws += align_up((size_t)B * F * TP, 64);
    float* rowsum = ws;                       ws += align_up((size_t)B * F * 4, 64);
    float* tpart = ws;
    // forward recompute (staged kernels = the reference graph)
    rc = leaf_gabor_conv_f32(x, B, T, kernel, F, K, y, taps, (size_t)2 * F * K * 4, stream);
    if (rc != LEAF_OK) return rc;
    rc = leaf_squared_modulus_f32(y, B, F, T, e, stream);
    if (rc != LEAF_OK) return rc;
    rc = leaf_gaussian_lowpass_f32(e, B, F, T, pool_w, pool_b, K, hop, raw, g, (size_t)F * K * 4, stream);
    if (rc != LEAF_OK) return rc;
    // PCEN + floor backward
    hipLaunchKernelGGL(pcen_bwd_scan_kernel, dim3(ceil_div(B * F, 4)), dim3(256), 0, st, raw, grad_out, B * F, F, TP, alpha,
                       delta, root, ema_w, 1e-12f, mode, ema, gpre, rowsum, (const int*)nullptr, 0, (float*)nullptr);
    LEAF_LAUNCH_CHECK();
    // pooling backward: window gradient needs e, sample gradient turns y into dy in place
    hipLaunchKernelGGL(pool_bwd_dg_kernel, dim3(ceil_div(K, 128), F), dim3(128), 0, st, e, gpre, B, F, T, TP, K, hop, padL, dg);
    LEAF_LAUNCH_CHECK();
    hipLaunchKernelGGL(param_reduce_kernel, dim3(F), dim3(kParamRedThreads), 0, st, gpre, dg, g, rowsum, pool_w, B, F, TP, K, mode,
                       (const float*)nullptr, 0, 0, (const int*)nullptr, g_pool_w, g_pool_b, g_alpha, g_delta, g_root, g_ema_w);
    LEAF_LAUNCH_CHECK();
    hipLaunchKernelGGL(pool_bwd_dy_kernel, dim3(ceil_div(T, 256), F, B), dim3(256), 0, st, y, g, gpre, F, T, TP, K, hop, padL);
    LEAF_LAUNCH_CHECK();
    // filterbank backward
    hipLaunchKernelGGL(dtaps_partial_kernel, dim3(ceil_div(K, 128), 2 * F, B), dim3(128), 0, st, y, x, T, 2 * F, K, padL,
                       tpart);
    LEAF_LAUNCH_CHECK();
    hipLaunchKernelGGL(dkernel_kernel, dim3(F), dim3(256), 0, st, tpart, taps, kernel, B, F, K, gabor_bounds(K), g_kernel);
    LEAF_LAUNCH_CHECK();
    if (g_x) {
        hipLaunchKernelGGL(dx_kernel, dim3(ceil_div(T, 256), B), dim3(256), 0, st, y, taps, T, 2 * F, K, padL, g_x);
        LEAF_LAUNCH_CHECK();
    }
    return LEAF_OK;
}

}  // extern "C"
