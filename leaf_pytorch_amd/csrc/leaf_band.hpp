// leaf_band.hpp -- band-limited filter tasks of the static workgroup forward kernel (leaf_fft_wg.hpp), round 5
// Part of libleaf_hip.so (gfx950 only); included by leaf_fft_wg.hpp.
//
// Why.  The workgroup kernel runs one 2048-point inverse transform per (block, filter).  Most Gabor filters occupy a narrow
// band: the spectrum R_f of the K-tap filter (impulse_responses.py:5-16) is a Gaussian of sigma_k = N / (2 pi sigma_f) bins
// around bin mu_f N / 2 pi, plus the side lobes of the truncation to K taps.  When all but eps^2 of its energy sits in a
// window of M = 256 or 512 bins, the M-point inverse transform of those bins gives y_f at every D-th sample (D = N / M) up
// to a phase that |.|^2 removes, and eight (M = 256) or four (M = 512) filters share the registers one 2048-point transform
// needs.  |y|^2 is pooled at the decimated rate with the window G~(tau) = D (g_f * phi_D)(tau), phi_D a low-pass with
// cutoff 1 / (2 D) (tools/gen_band_phi.py), which represents sum_n g[n] e[n] exactly wherever e is band-limited to what the
// decimated grid resolves; block boundaries cut the DECIMATED sequence (an exact partition of that sum), and the frames
// whose window is cut by the clip's ends ("edge frames": the reference zero-pads the energies, frontend.py:15-19 ->
// pooling.py:31-42) take dense per-block tables W~ = D (W * phi_D) that are exact for the block's periodic signal.
// tools/band_proto.py is the fp64 model of all of this; it measures <= 4e-7 of a filter's largest pooled value at the
// default initialisation and <= 7e-6 over random (mu, sigma, pooling width) with the decision rule below.
//
// Decision (per filter, per call, on the device, from the table the call has just built -- fft_prep_band_kernel):
//     class c (M = 256 c) is admissible when, with the window [kb, kb + M) placed around the filter's centre bin inside 1..1024
//     (round 6, forward launches of the 2048-sample plan: inside 1..1151 -- the window may cross Nyquist, kWgFwdBins),
//         sum_{k outside} R^2 <= eps^2 sum R^2        (what the short transform drops; eps = 3e-6)
//         sum_i |R_i R_{i + d}| <= eta sum R^2, d = M/2, 3M/4   (content of |y|^2 the decimated grid would alias; eta = 2e-4 in round 5,
//                                                                 1e-6 since round 6 -- kBandEtaFree, and the comment at kBandAliasEdge)
//     or (round 6) when the pooling bias of the call is at least the minimal bias band_need derives from the same sums.
// Filters that fail both classes keep the 2048-point task, so a call with wide filters costs what it did before.
//
// Layout of a band task (A = 16: M = 256 = 16 x 16, G = 8 filters; A = 32: M = 512 = 32 x 16, G = 4 filters; D = G = 128 / A):
//   phase 1  lane = g (A/2) + c: registers (h, r) = bin j = j1 + A j2 of filter g's window, j1 = c + (A/2) h, j2 = r;
//            spectral multiply fused with the first DIT stage, then the 16-point transform over j2 (register i <-> m2 = brev4(i)),
//            twiddle W_M^(j1 m2) from the kernel's 2048-point table, and the transposition through the wave's scratch (rows = registers,
//            columns = lanes: the same conflict-free add-tid stores as the 2048-point transform);
//   phase 2  lane = l2 G + g, l2 = m2 (A = 32) or m2 mod 8 (A = 16: m2 = l2 + 8 h): the A-point transform over j1 (two 16-point
//            or one 32-point), register <-> m1: decimated sample m = 16 m1 + m2, time n_c + D m;
//   pooling  64-sample rows rho = m1 (A = 32) or 2 m1 + h (A = 16), position D l2 inside the row: the weight of (row, frame)
//            is G~(64 rho - is(frame) + D l2), one of NV = 21 / 17 per-lane vectors, as in the 2048-point task;
//   sums     halving butterfly over the lanes of a filter (lane bits above log2 G), ds_add_f32 into the clip's LDS sums.
#pragma once
#include <utility>
#include <hip/hip_fp16.h>
#include "leaf_fft.hpp"

namespace {

#ifndef LEAF_BAND_EDGE_PIPE
#define LEAF_BAND_EDGE_PIPE 0      // band tasks: 1 = the next edge frame's table requested before the current one is consumed -- measured 7 % SLOWER
#endif                             // (0.1314 vs 0.1223 ms at cfg1, same box: 32 more live registers, 25 more spill instructions per task)
#ifndef LEAF_BAND_EDGE_EARLY
#define LEAF_BAND_EDGE_EARLY 0     // band tasks: 1 = the first edge table requested before the reduction of the regular frames -- 2 % slower
#endif                             // (0.1251 vs 0.1222 ms at cfg1, same box) and 32 B of scratch; 0: after it, no scratch
#ifndef LEAF_PREP_ABLATE
#define LEAF_PREP_ABLATE 0         // measurement only (results wrong): fft_prep_band_kernel without 1 = the edge-table workgroups, 2 = the G~ workgroup, 4 = the decision sums, 8 = the first-block spectra, 16 = the taps' transform, 32 = the twiddle tables (bits)
#endif
#ifndef LEAF_BAND_PW_EARLY
#define LEAF_BAND_PW_EARLY 1       // band tasks: pooling weights requested before the second transforms (0: after them, A/B)
#endif
constexpr int kBandLh = 12;                                   // half length of phi_D in decimated samples (leaf_band_phi.inc)
constexpr float kBandEps2 = 9e-12f;                           // eps^2, eps = 3e-6
constexpr float kBandEta = 2e-4f;
// Round 6: the energy a band window may drop follows the filter's pooling BIAS -- where the pooling window low-passes the cross terms.
// With y = y_W + y_o (inside / outside the window), sum g |y|^2 - sum g |y_W|^2 = sum g |y_o|^2 + 2 Re sum g y_W conj(y_o).
//   * Quadratic term.  For |x| <= 1 the worst input is one full-scale tone whose two components (amplitude 1/2 each) both fall on the
//     largest dropped bin: sum g |y_o|^2 <= G_0 max_{k outside} R_k^2 / 2.  A pooled value is p = bias_f + sum g |y|^2 >= bias_f
//     (pooling.py:31-42), so the class costs at most kBandQuadTol = 5e-6 of ANY output wherever
//         bias_f >= bmin = G_0 max_{k outside} R_k^2 / (2 kBandQuadTol)
//     (measured, profiles/r06/bias_bound_check.txt: a tone on a dropped side lobe reaches half of this bound -- one of its two
//     components).  Round 5's rule -- all but eps^2 = 9e-12 of the ENERGY inside the window, whatever the bias -- stays as it is
//     (`strict`): it covers biases down to the floor 1e-5 (frontend.py:84).
//   * Cross term: FIRST order in the dropped amplitude; eps = 3e-6 kept it at 2 eps whatever the pooling window.  A class that
//     drops more needs the pooling window's own low-pass: a strong component in the filter's core (within 2 sigma_k of the centre
//     bin) and a dropped one are >= dmin bins apart, where the window's spectrum is down to
//         gamma = exp(-(w sigma_p)^2 / 2) + 2 g[edge] / (w G_0),   w = 2 pi dmin / N
//     (its Gaussian main lobe and the 1/w side lobes of its truncation at the K taps); the class is admissible beyond the strict
//     rule only if 2 gamma sqrt(sum_{outside} R^2) / R_peak <= kBandCrossMax = 6e-6, round 5's own level.  One-sample pooling
//     windows (gamma = 1) therefore decide exactly as in round 5.
// The default bias 1.0 and pooling width 0.4 admit the four sigma = 48 filters of the 16 kHz default initialisation (bmin ~ 0.4)
// and the 23 sigma = 96 filters of the 32 kHz one; the 16 kHz filter next to Nyquist (its main lobe's tail is what the window drops:
// bmin ~ 15) and the sigma = 64 / 192 ones (cross term) stay on full transforms.  The prep kernels record bmin per filter and class as an fp16 rounded up (+inf: the aliasing or the cross-term criterion
// fails), band_build_plan compares it with THIS call's bias: the tables (also the frozen-parameter ones) do not depend on the bias.
// The backward's band tasks take the same decision (measured: gradients stay at ~1e-6 of their column's largest entry,
// profiles/r06/exp_bwd_bias.txt); LEAF_ALGO_STRICT_BAND_CLASSES / LEAF_FLAG_BWD_STRICT_BAND_CLASSES take the strict rule alone (round 5's decision).
constexpr float kBandQuadTol = 5e-6f;
constexpr float kBandCrossMax = 6e-6f;                        // 2 eps of round 5: the cross term a wider class may add, relative to the frame's energy
constexpr int kBandNever = 0x7C00;                            // fp16 +inf
// Backward (round 6).  A band task of the backward turns v = 2 de conj(z) (de: the pooling's gradient at the decimated rate) into the
// M-point spectrum V and takes d mu, d sigma as dot products of conj(A') V with R_mu, R_sigma on the window's bins.  V is the filter's
// band WIDENED by the spectrum of the pooling window (its main lobe and the 1/w side lobes of its truncation); what passes the window's
// edge wraps around to the other edge, where it meets whatever R_mu / R_sigma -- wider than R itself: R_sigma ~ ((k - k0)^2 / sigma_k^2 - 1) R
// -- still have there.  At the lower-sigma end of a class (sigma = 7.5 .. 8 on 512 points, 15 .. 16 on 256) that product put d sigma off by
// 1e-4 .. 5e-4 of itself and up to 1.5e-4 of the column's largest entry (profiles/r06/bwd_derivative_spectra.txt; found by the per-column
// gradient metric on extended fuzz seeds -- round 5's metric and seeds did not see it).  The backward's classes therefore also ask that the
// main lobe of the derivative spectra -- kBandDerivCore sigma_k either side of the centre bin, where x^2 exp(-x^2 / 2) is down to 1e-5 --
// end an eighth of the window's length before the nearer edge (the margin the wrapped part lands in).  Geometric on purpose: the far side
// lobes of a truncated filter's derivative spectra only enter at second order, and an energy bound on them would switch every truncated
// filter's band task off (tried: cfg2's backward 3.1 -> 5.8 ms).
constexpr float kBandDerivCore = 5.6f;
constexpr int kBandDerivMarginDiv = 8;
__device__ __forceinline__ bool band_deriv_fits(int k0, int kb, int M, float sk) {
    return (float)min(k0 - (kb - 1), kb + M - k0) >= kBandDerivCore * sk + (float)(M / kBandDerivMarginDiv);
}
// gamma of the comment above: pooling width s (clamped, impulse_responses.py:75), window length K, distance dmin bins of N
__device__ __forceinline__ float band_pool_gamma(float s, int K, float dmin, int N) {
    if (!(dmin >= 8.0f)) return 1.0f;
    const float sp = s * 0.5f * (float)(K - 1), w = 6.2831853f * dmin / (float)N;
    const float g0 = 0.95f * 2.5066283f * sp;                 // sum g >= 0.95 sqrt(2 pi) sigma_p for s <= 0.5
    return fminf(1.0f, __expf(-0.5f * (w * sp) * (w * sp)) + 2.0f * __expf(-0.5f / (s * s)) / (w * g0));
}
// bmin of a class as an fp16 code (rounded up).  out2 / ac_* / tot: the sums of the decision over the table (which has 1 / N folded
// in); mx: the largest dropped R^2 of the table; rpk: |R| at the centre bin; gam: band_pool_gamma; s: the clamped pooling width.
// Two refinements the seeded fuzz forced (profiles/r06/bias_bound_fuzz_cases.txt):
//   * a window that ends at DC or Nyquist drops the IMAGES of components it keeps (x is real): kept and dropped part then beat at
//     twice the component's distance from the edge, which no pooling window removes -- measured 2.4x the quadratic term alone for a
//     filter whose main lobe reaches DC; the quadratic bound is taken kBandAdjacent = 6 times;
//   * the cross term against a WEAK kept component: with amplitudes A1 (in the core) and A2 (dropped), the error relative to
//     p = b + G_0 (A1 R_pk)^2 / 4 peaks at A1^2 = 4 b / (G_0 R_pk^2): gamma (R_o / R_pk) sqrt(G_0 / b) / 2 -- it needs
//     b >= G_0 (gamma R_o / R_pk)^2 / (4 kBandCrossTol^2)   (measured: 7e-5 at b = 0.02 with a 9-sample pooling window).
constexpr float kBandAdjacent = 6.0f;
// ... and 60 times on what the window drops AT DC -- bins 0, -1 .. -63, the filter's lower tail where it reaches DC: a DC offset (or any step: a clip
// that starts away from zero) excites the filter's in-band transient AND the bins next to DC coherently, and their cross term is linear in R(0), not
// quadratic -- measured on a filter 4.5 sigma_k above DC under x = 0.82 + a weak tone: 4e-5 of (bias 0.3 + pooled energy) where the factor 6 promised
// 2.6e-6 (tools/dbg_dc_edge.py; case 117 of the fp64 model's fuzz, tools/band_proto.py --bias-fuzz).  The same weight goes on the dropped energy at DC
// in the cross-term bounds and in the bias-free energy bound.  Chosen on the fp64 model's hunt (profiles/r06/bias_rule_fp64_model.txt, 20 000+ cases):
// 30 leaves several cases at 2.1e-5, 60 one (DC offset under a filter 4.7 sigma_k above DC, bias 0.3: 2.1e-5; the rest <= 1.3e-5), 100 none -- but 100 costs the
// default bank its filter 6 and 300 two of them (truncation side lobes of sigma = 48 sit on the DC bins), 4 .. 10 % of the headline for a case inside the north star by 5x.
constexpr float kBandAdjacentDC = 60.0f;
constexpr float kBandCrossTol = 5e-6f;
constexpr float kBandEtaWide = 1e-5f;
// Round 6, found with windows that cross Nyquist and then on ordinary ones (profiles/r06/band_alias_pairs.txt): two spectral lines inside
// a window more than ~0.3 M bins apart beat in |y|^2 near or above the decimated grid's Nyquist, where the interpolation kernel phi_D is
// in its transition band -- on an edge frame (a pooling window cut by the clip's end does not low-pass it) and, weaker, on regular ones.
// Round 5's eta = 2e-4 admitted sigma = 15 .. 16 to 256 points, where two tones of amplitude 0.5 at +- 60 bins of the centre are off by
// 1.5e-4 of (bias 0.1 + pooled energy); a window centred on Nyquist holds every line TOGETHER with its mirror image, so ONE full-scale tone
// does the same.  Now: the pair sums may reach kBandEtaWide of the filter's energy only under a minimal bias (below), and kBandEtaFree = 1e-6
// without one (at 1e-5 bias-free the window-built fuzz still found 2.6 .. 4.1e-5 at biases of 0.02 .. 0.05; round 5's 2e-4 stays under
// LEAF_ALGO_STRICT_BAND_CLASSES).  The pair sum: max over the lags M / 2, 3 M / 4 of sum R_i R_(i+lag), + the mirrored pairs R_k R_(2048-k),
// 2 (k - 1024) >= 5 M / 16, of a window across Nyquist at a quarter of that weight (measured 0.13).
constexpr float kBandEtaFree = 1e-6f;       // the aliasing bound of the BIAS-FREE part of the decision (a.eta of default launches): between it and kBandEtaWide the pair-sum bias decides
                                            // (2e-6 at first: the fp64 model's hunt had a pair 0.45 M apart at 2.5e-5 under a bias of 0.02; 1e-7 sits on the fp32 noise of the table: peak x transform noise summed over the main lobe flipped a sigma = 43 filter of the 32 kHz bank)
// Two parts, both per unit of the pair sum in true units (N^2 sum ...), w = pi M / N the beat's frequency in radians per sample:
//   edge frames: the cut window passes 1 / w of it whatever its width -- measured 2.7e-3 / w (pool_w 0.5, first frame), doubled;
//   regular frames: the Gaussian main lobe exp(-(w sigma_p)^2 / 2) of the pooling window, i.e. only windows a few samples wide: 0.05 G_0 per
//   unit for a one-sample window (the beat itself), doubled.
constexpr float kBandAliasEdge = 5.4e-3f;
constexpr float kBandAliasReg = 0.1f;
constexpr float kBandAliasTol = 1e-5f;
constexpr float kBandMirrorW = 0.25f;
__device__ __forceinline__ int band_need(float out2, float mx, float ac_a, float ac_b, float tot, float eta, float rpk, float gam, float s, int K, int N,
                                         float pm, int M, float outdc, float mxdc) {
    // (the aliasing criterion at a twentieth of eta: the truncation side lobes INSIDE the window of these filters put more of |y|^2
    // at the decimated grid's Nyquist than a filter that passes the strict rule does -- sigma = 54.6 under a 9-sample pooling
    // window: 2.6e-5; the admitted default filters are at 3 .. 6e-6 of their energy)
    if (!(ac_a <= kBandEtaWide * tot && ac_b <= kBandEtaWide * tot)) return kBandNever;
    // (the DC weight of kBandAdjacentDC: what the window drops on bins 0, -1 .. -63 counts kBandAdjacentDC / kBandAdjacent times)
    const float wdc = kBandAdjacentDC / kBandAdjacent - 1.0f;
    out2 += wdc * outdc;
    if (!(2.0f * gam * sqrtf(out2) <= kBandCrossMax * rpk)) return kBandNever;      // equal amplitudes: round 5's level whatever the bias
    const float g0 = 2.5066283f * s * 0.5f * (float)(K - 1);                       // sum g <= sqrt(2 pi) sigma_p
    const float bq = fmaxf(kBandAdjacent * mx, kBandAdjacentDC * mxdc) * g0 * (float)N * (float)N / (2.0f * kBandQuadTol);
    const float ro = gam * sqrtf(out2) / rpk;
    const float bc = g0 * ro * ro / (4.0f * kBandCrossTol * kBandCrossTol);          // (out2 carries the DC weight: a DC offset beside a weak core)
    const float wl = 3.14159265f * (float)M / (float)N, wsp = wl * s * 0.5f * (float)(K - 1);
    const float ba = (fmaxf(ac_a, ac_b) + kBandMirrorW * pm) * (float)N * (float)N * (kBandAliasEdge / wl + kBandAliasReg * g0 * __expf(-0.5f * wsp * wsp)) / kBandAliasTol;
    const float bmin = fmaxf(fmaxf(bq, bc), ba);
    if (!(bmin < 60000.0f)) return kBandNever;
    return (int)__half_as_ushort(__float2half_ru(fmaxf(bmin, 6.2e-5f)));          // (>= the smallest normal fp16)
}
// does this call's bias admit the class beyond the strict rule?  (bias = NULL or relax off: no; NaN bias: no)
__device__ __forceinline__ bool band_bias_admits(const float* __restrict__ bias, int f, bool relax, int code) {
    if (!bias || !relax || (code & 0xFFFF) == kBandNever) return false;
    return bias[f] >= __half2float(__ushort_as_half((unsigned short)(code & 0xFFFF)));
}

constexpr int kBandMaxFilters = 256;                          // the plan lives in LDS

__host__ __device__ constexpr int brev4(int i) { return ((i & 1) << 3) | ((i & 2) << 1) | ((i & 4) >> 1) | ((i & 8) >> 3); }
__host__ __device__ constexpr int band_d(int A) { return 128 / A; }           // decimation = filters per task
__host__ __device__ constexpr int band_lpf(int A) { return A / 2; }           // lanes per filter
__host__ __device__ constexpr int band_m(int A) { return 16 * A; }            // transform length
__host__ __device__ constexpr int band_lphi(int A) { return kBandLh * band_d(A); }
__host__ __device__ constexpr int band_gcd(int a, int b) { return b == 0 ? a : band_gcd(b, a % b); }
// (D given explicitly: the 4096-sample plan runs the A = 32 layout at D = 8 -- four filters per task, 128-sample rows)
__host__ __device__ constexpr int band_rl(int A, int D) { return A / 2 * D; }  // samples a register row spans
// the per-lane weight vectors of the decimated pooling: c0 = RL rho - is(frame) runs over c0min + PG k, k < NV
__host__ __device__ constexpr int band_c0min_d(int K, int hop, int A, int D) {
    const int padl = K / 2 + K % 2 - 1, rl = band_rl(A, D), pg = band_gcd(rl, hop), lo = -kBandLh * D - rl + D;
    return lo + (((padl - lo) % pg) + pg) % pg;
}
__host__ __device__ constexpr int band_nv_d(int K, int hop, int A, int D) {
    return (K - 1 + kBandLh * D - band_c0min_d(K, hop, A, D)) / band_gcd(band_rl(A, D), hop) + 1;
}
__host__ __device__ constexpr int band_gz_len_d(int K, int hop, int A, int D) {      // table entries tau = c0min + D j
    return band_gcd(band_rl(A, D), hop) / D * (band_nv_d(K, hop, A, D) - 1) + band_rl(A, D) / D;
}
__host__ __device__ constexpr int band_c0min(int K, int hop, int A) { return band_c0min_d(K, hop, A, band_d(A)); }
__host__ __device__ constexpr int band_nv(int K, int hop, int A) { return band_nv_d(K, hop, A, band_d(A)); }
__host__ __device__ constexpr int band_gz_len(int K, int hop, int A) { return band_gz_len_d(K, hop, A, band_d(A)); }
__host__ __device__ constexpr int band_gz_floats(int K, int hop) { return (band_gz_len(K, hop, 16) + band_gz_len(K, hop, 32) + 3) / 4 * 4; }
// geometries the band tasks are built for: static odd windows whose hop and block length the decimations divide and
// whose frame range per block the widened windows do not change
__host__ __device__ constexpr bool band_geometry_ok(int K, int hop) {
    // (K = 801 / hop = 320 on 2048-sample blocks passes the checks below and runs correctly -- 1.4e-6 against the oracle -- but at
    // the 32 kHz default initialisation 49 of 80 filters are truncated too hard for a band class and the 4096-sample kernel
    // stays ahead: 2.24 ms with band tasks on 2048-sample blocks against 2.33 ms, profiles/r05/exp_cfg2_2k.txt.  Not enabled.)
    if (!(K == 401 && hop == 160)) return false;
    const int padl = K / 2 + K % 2 - 1, ls = fft_block_len(K, hop, true), lphi = band_lphi(16);
    const int dmin = -((K - 1 - padl) / hop), dmax = (ls - 1 + padl) / hop;
    const int dmin_w = -((K - 1 - padl + lphi) / hop), dmax_w = (ls - 1 + padl + lphi) / hop;
    return hop % 8 == 0 && ls % 64 == 0 && band_gcd(64, hop) % 8 == 0 && dmin == dmin_w && dmax == dmax_w && dmax - dmin + 1 <= 16;
}
// LDS of the band area: plan header (4) | edge list (4 kBandMaxEdge) | task descriptors (F + 4) | members (F + 16) |
// three class lists (3 F) | the packed records (F).  (The twiddles W_M^(j1 m2) of both classes are entries of the 2048-point table the kernel has
// anyway: W_M^(j1 m2) = W_2048^(l k1) with k1 = 2 m2 and l = 2 j1 (M = 512) or 4 j1 (M = 256).  Their own tables cost 6 KB,
// which the streaming-finalize kernels do not have: BASELINE configs[3] / [4] fell back to partial sums in HBM, 2.7x traffic.)
constexpr int kBandPlanHead = 4 + 4 * kBandMaxEdge;
__host__ __device__ constexpr int band_lds_ints(int F) { return (kBandPlanHead + (F + 4) + (F + 16) + 4 * F + 3) / 4 * 4; }
__host__ __device__ constexpr size_t band_lds_bytes(int F) { return (size_t)band_lds_ints(F) * 4; }
constexpr int kBandInvalid = 1 << 30;                         // member entry: padding of a partly filled task

template <bool HALF, bool SKIP1>
__device__ __forceinline__ void fft2048w(float (&re)[32], float (&im)[32], float* scr, unsigned scr_lds, const float2* twl,
                                         const float2* twh, int lane);   // leaf_fft_wg.hpp (which includes this header above it)
__device__ __forceinline__ void fft_build_twiddles_wg(float2* twl, float2* twp, int tid, int nthreads);

#ifndef LEAF_INST_TU               // non-template kernel: compiled once, in leaf_kernels.hip
#include "leaf_band_phi.inc"
static_assert(kBandPhiLh == kBandLh, "leaf_band_phi.inc was generated for another filter length");

struct BandTabArgs {
    int T, L, hop, padL;
    float eps2, eta;
    int force;             // tests / tools: 0 decide, 1 / 2 every filter in the 256- / 512-point class, 3 none
    int edge_only;         // frozen-parameter tables (leaf_forward_prepared_f32): grid (F, n_edge), only the edge tables of this clip length
    int* rec;
    float* gz;
    float* edge;
    float* gz2;            // (backward) the same two tables for the window g_f[j] (j - c)^2, c = (K - 1) / 2: d pool_w at the decimated
    float* edge2;          // rate (impulse_responses.py:74-80: d g / d s = g (j - c)^2 / (c^2 s^3)); NULL: not built
    int* elist;
    int* classes;          // leaf_band_classes_f32: [F] the transform length each filter gets (NULL: not asked for)
    const float* cls_bias; // ... for these pooling biases (round 6: the energy bound follows the bias; NULL: the strict decision)
    int n_edge;
    BandEdge e[kBandMaxEdge];
    // the main kernel's first blocks (round 5): waves 1..7 of the workgroups (f, 0) -- idle while wave 0 transforms the filter's
    // taps -- transform the first block of main-kernel workgroups w = 7 f + wave - 1 (+ 7 F, ...) < G into spec0[w]: the one
    // forward transform nothing in the main kernel can overlap with.  Their transposition scratch is the launch's dynamic LDS.
    const void* x;
    int io_bf16, B, nblk, G;
    float2* spec0;
    int cross;             // != 0 (forward launches of the 2048-sample plan, round 6): windows may reach bin kWgFwdBins - 1 (beyond Nyquist)
    int bwd_slabs;         // (backward) != 0: the class decision also asks band_deriv_fits; 1 (2048-sample plan): two more grid rows, (f, 2 + n_edge) and (f, 3 + n_edge), build the spectra of d w / d mu and
                           // d w / d sigma into slabs 1 and 2 of H (what fft_prep_kernel's grid (F, 3) does): one table launch
};

// sum over a 16-lane row (every lane of the row gets it)
__device__ __forceinline__ float band_row_sum(float v) {
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0xb1, 0xf, 0xf, false));    // quad_perm [1,0,3,2]
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x4e, 0xf, 0xf, false));    // quad_perm [2,3,0,1]
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x124, 0xf, 0xf, false));   // row_ror:4
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x128, 0xf, 0xf, false));   // row_ror:8
    return v;
}

// fft_prep_kernel (real-spectrum form, forward tables) + the tables of the band tasks.  Grid (F, 2 + n_edge):
//   workgroup (f, 0): the tables of fft_prep_kernel (wave 0 runs the filter's 2048-point transform), then all waves take the
//                     seven sums of the class decision from the spectrum wave 0 left in LDS;
//   workgroup (f, 1 + s): the edge table of edge entry s, both classes   } on CUs the F transform workgroups leave idle,
//   workgroup (f, 1 + n_edge): the decimated pooling windows G~ of both classes } and done before wave 0's transform is
// Sixteen lanes per table entry: the sum over the window samples within lphi of the entry's position.
// (Workgroup (0, 0) also copies the edge list to device memory for the main kernel.)
__global__ __launch_bounds__(kPrepWaves * 64) void fft_prep_band_kernel(const float* __restrict__ kernel, const float* __restrict__ pool_w,
                                                                        int F, int K, int GZ, GaborBounds bd, float2* __restrict__ H,
                                                                        float* __restrict__ Gz, int* __restrict__ col_of, const BandTabArgs a) {
    __shared__ float2 s_twl[32 * 64];
    __shared__ float2 s_twh[64];
    __shared__ float2 s_twp[64];
    __shared__ float s_scr[32 * 65];
    __shared__ float2 s_taps[kFftN / 2 + 64];
    __shared__ float Rs[kFftN];
    __shared__ float gs[64 * kPoolRowsMax];
    __shared__ float phis[2][kBandLh * 8 + 1];               // phi_8 | phi_4 (one half each)
    __shared__ float red[8][14];
    __shared__ int es[kBandMaxEdge][4];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, f = blockIdx.x;
    const int l16 = lane & 15;
    // one table entry per 16-lane row: value = sum over pp = lo .. hi of gs[pp + goff] phi[|p0 - pp|]
    auto entry = [&](const float* phi, int p0, int lo, int hi, int goff, const float* win) {
        float acc = 0.0f;
#pragma unroll 4
        for (int pp = lo + l16; pp <= hi; pp += 16) {
            const int u = p0 - pp;
            acc = fmaf(win[pp + goff], phi[u < 0 ? -u : u], acc);
        }
        return band_row_sum(acc);
    };
    float* gs2 = gs + 64 * kPoolRowsMax / 2;                              // (backward tables) g_f[j] (j - c)^2; K <= 640 here
    // both windows in one pass (the backward's tables: the same taps, the same phi)
    auto entry2 = [&](const float* phi, int p0, int lo, int hi, int goff, float& v2) {
        float acc = 0.0f, acc2 = 0.0f;
#pragma unroll 4
        for (int pp = lo + l16; pp <= hi; pp += 16) {
            const int u = p0 - pp;
            const float ph = phi[u < 0 ? -u : u];
            acc = fmaf(gs[pp + goff], ph, acc);
            acc2 = fmaf(gs2[pp + goff], ph, acc2);
        }
        v2 = band_row_sum(acc2);
        return band_row_sum(acc);
    };
    if (tid <= kBandLh * 8) phis[0][tid] = kBandPhi8[tid];
    else if (tid >= 128 && tid - 128 <= kBandLh * 4) phis[1][tid - 128] = kBandPhi4[tid - 128];
#pragma unroll
    for (int s = 0; s < kBandMaxEdge; ++s)
        if (tid == 256 + s) { es[s][0] = a.e[s].c; es[s][1] = a.e[s].m; es[s][2] = a.e[s].lo; es[s][3] = a.e[s].hi; }
    if (a.bwd_slabs && (int)blockIdx.y >= 2 + a.n_edge) {
        // ---- (backward) the spectrum of d w / d mu or d w / d sigma: fft_prep_kernel's workgroup (f, which)
        const int which = (int)blockIdx.y - (1 + a.n_edge);
        fft_prep_front(kernel, pool_w, F, K, GZ, bd, Gz, f, which, s_twl, s_twh, s_taps, nullptr, tid);
        __syncthreads();
        if (wave == 0) fft_prep_transform(F, K, 1, H, col_of, nullptr, f, which, s_twl, s_twh, s_scr, s_taps, nullptr, lane);
        return;
    }
    if ((LEAF_PREP_ABLATE & 1) && blockIdx.y > 0 && (int)blockIdx.y < 1 + a.n_edge) return;
    if ((LEAF_PREP_ABLATE & 2) && (int)blockIdx.y == 1 + a.n_edge) return;
    if (blockIdx.y > 0 || a.edge_only) {
        // ---- edge table W~[m] of entry s, both classes, in the register order of the class (band_task: the lane reads entry
        // k LPF + l2 of its register k): the window's samples [pa, pb) relative to the block, and the image of m D within lphi of them
        const float half = 0.5f * (float)(K - 1);
        for (int j = tid; j < K; j += kPrepWaves * 64) {                   // the pooling window: as fft_prep_front evaluates it, or
            if (a.edge_only) {                                             // (frozen-parameter tables) as it left it in the pooling row
                gs[j] = Gz[(size_t)f * GZ + kGPad + j];
            } else {
                const float q = ((float)j - half) / (pool_sigma(pool_w[f], K) * half);
                gs[j] = expf(-0.5f * (q * q));
            }
            const float tj = (float)j - half;
            gs2[j] = gs[j] * (tj * tj);
        }
        __syncthreads();
        const int grp = tid >> 4;
        if (!a.edge_only && (int)blockIdx.y == 1 + a.n_edge) {
            // ---- decimated pooling windows G~(tau) = D sum_u g[tau - u] phi_D[|u|], tau = c0min + D j, of both classes
            const int len16 = band_gz_len(K, a.hop, 16), len32 = band_gz_len(K, a.hop, 32);
            const int gzs = band_gz_floats(K, a.hop);                       // (run-time gcd loops: evaluated ONCE, not per table entry)
            float* gzf = a.gz + (size_t)f * gzs;
            float* gz2f = a.gz2 ? a.gz2 + (size_t)f * gzs : nullptr;
#pragma unroll
            for (int cls = 0; cls < 2; ++cls) {
                const int A = 16 << cls, D = band_d(A), lphi = band_lphi(A), c0 = band_c0min(K, a.hop, A), len = cls ? len32 : len16;
                for (int j = grp; j < len; j += kPrepWaves * 4) {
                    const int tau = c0 + D * j;
                    if (a.gz2) {
                        float v2;
                        const float v = entry2(phis[cls], tau, max(0, tau - lphi), min(K - 1, tau + lphi), 0, v2);
                        if (l16 == 0) {
                            gzf[(cls ? len16 : 0) + j] = (float)D * v;
                            gz2f[(cls ? len16 : 0) + j] = (float)D * v2;
                        }
                    } else {
                        const float v = entry(phis[cls], tau, max(0, tau - lphi), min(K - 1, tau + lphi), 0, gs);
                        if (l16 == 0) gzf[(cls ? len16 : 0) + j] = (float)D * v;
                    }
                }
            }
            return;
        }
        const int s = a.edge_only ? blockIdx.y : blockIdx.y - 1;
        if (f == 0 && s == 0 && tid < 4 * kBandMaxEdge) a.elist[tid] = es[tid >> 2][tid & 3];   // (edge_only: nobody else does)
        const int c = es[s][0], goff = c * a.L - (es[s][1] * a.hop - a.padL);
        const int pa = es[s][2] - c * a.L, pb = es[s][3] - c * a.L;
#pragma unroll
        for (int cls = 0; cls < 2; ++cls) {
            const int A = 16 << cls, D = band_d(A), lphi = band_lphi(A), M = band_m(A);
            float* tab = a.edge + (((size_t)f * 2 + cls) * kBandMaxEdge + s) * 512;
            for (int m = grp; m < M; m += kPrepWaves * 4) {
                int p0 = m * D;
                if (p0 - kFftN + lphi >= pa) p0 -= kFftN;
                else if (p0 + kFftN - lphi < pb) p0 += kFftN;
                const int m1 = m >> 4, m2 = m & 15;
                const int idx = cls ? brev5(m1) * 16 + m2 : (16 * (m2 >> 3) + brev4(m1)) * 8 + (m2 & 7);
                if (a.edge2) {
                    float v2;
                    const float v = entry2(phis[cls], p0, max(pa, p0 - lphi), min(pb - 1, p0 + lphi), goff, v2);
                    if (l16 == 0) { tab[idx] = (float)D * v; a.edge2[(tab - a.edge) + idx] = (float)D * v2; }
                } else {
                    const float v = entry(phis[cls], p0, max(pa, p0 - lphi), min(pb - 1, p0 + lphi), goff, gs);
                    if (l16 == 0) tab[idx] = (float)D * v;
                }
            }
        }
        return;
    }
    fft_prep_front(kernel, pool_w, F, K, GZ, bd, Gz, f, 0, s_twl, s_twh, s_taps, gs, tid);
    if (a.spec0 && tid < 64) {                                             // the half-wave twiddles in fft2048w's layout (fft_build_twiddles_wg)
        const int h = tid >> 5, e = tid & 31, j = (e >> 1) + 16 * (e & 1);
        float sn, cs;
        sincospif(2.0f * (float)j / 64.0f, &sn, &cs);
        s_twp[tid] = h ? make_float2(cs, -sn) : make_float2(1.0f, 0.0f);
    }
    __syncthreads();
    if (a.elist && f == 0 && tid >= 64 && tid < 64 + 4 * kBandMaxEdge) a.elist[tid - 64] = es[(tid - 64) >> 2][(tid - 64) & 3];
    if (wave == 0) fft_prep_transform(F, K, 1, H, col_of, nullptr, f, 0, s_twl, s_twh, s_scr, s_taps, Rs, lane);
    else if (a.spec0 && !(LEAF_PREP_ABLATE & 8)) {
        // ---- first blocks of the main kernel's workgroups, with the main kernel's own transform (fft2048w: the bits it would
        // compute itself; s_twl is the table both transforms share)
        extern __shared__ __attribute__((aligned(16))) float dyn_scr[];
        float* scr = dyn_scr + (size_t)(wave - 1) * kWgScrFloats;
        const unsigned scr_lds = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(__attribute__((address_space(3))) float*)scr);
        const OwnedClips deal{a.B * a.nblk, a.G, a.nblk};
        for (int w = f * (kPrepWaves - 1) + wave - 1; w < a.G; w += F * (kPrepWaves - 1)) {
            if (deal.count(w) <= 0) continue;
            const int gb = deal.start(w), b = gb / a.nblk, c = gb - b * a.nblk, n_c = c * a.L;
            const float* xb = static_cast<const float*>(a.x) + (size_t)b * a.T;
            const unsigned short* xh = static_cast<const unsigned short*>(a.x) + (size_t)b * a.T;
            float are[32], aim[32];
#pragma unroll
            for (int r = 0; r < 32; ++r) {
                const int i = 64 * r + lane;                              // block rotated left by padL samples (leaf_fft_wg_kernel)
                const int n = n_c - a.padL + ((i + a.padL) & (kFftN - 1));
                const int nc = min(max(n, 0), a.T - 1);
                const float v = a.io_bf16 ? __uint_as_float((unsigned)xh[nc] << 16) : xb[nc];
                are[r] = (n >= 0 && n < a.T) ? v : 0.0f;
                aim[r] = 0.0f;
            }
            fft2048w<false, false>(are, aim, scr, scr_lds, s_twl, s_twp, lane);
            float2* dst = a.spec0 + (size_t)w * kWgRingFwdFloat2;
#pragma unroll
            for (int i = 0; i < 32; ++i) {
                const int k = brev5(i);
                if (k < kWgFwdBins / 64) dst[64 * k + lane] = make_float2(are[i], aim[i]);   // bins 0..1151 (kWgFwdBins)
            }
        }
    }
    __syncthreads();
    if (LEAF_PREP_ABLATE & 4) return;
    // ---- the seven sums of the decision
    const float mu = fminf(fmaxf(kernel[2 * f], 0.0f), 3.14159274101257324f);
    const int k0 = (int)rintf(mu * (float)(kFftN / 6.283185307179586));
    float sums[7] = {0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f};   // tot | out2, ac(M/2), ac(3M/4) of class 1 | of class 2
    float dcs = 0.0f, mxdc = 0.0f;                                 // what every window drops at DC: bins 0, -1 .. -63 are entries 0 .. 63 (kBandAdjacentDC)
    float pmir[2] = {0.0f, 0.0f};                                  // mirrored pairs |R_k R_(2048-k)| inside a window that crosses Nyquist, 2 (k - 1024) >= 5 M / 16
    float mxo[2] = {0.0f, 0.0f};                                   // the largest dropped R^2 per class (round 6: the bias bound)
    int kbv[2];
#pragma unroll
    for (int cls = 0; cls < 2; ++cls) {
        const int M = 256 << cls;
        // bins kb .. kb + M - 1, centred on the filter.  Backward launches and LEAF_ALGO_STRICT_BAND_CLASSES keep the window inside the half
        // spectrum 1..1024 (a.cross = 0); the forward's ring holds bins 0..1151 (kWgFwdBins), so its windows may cross Nyquist -- a filter whose pass band
        // reaches beyond pi (the top one or two of the default mel bank) is then centred in its window instead of cut by it (round 6)
        const int kb = min(max(k0 - M / 2, 1), (a.cross ? kWgFwdBins : kFftN / 2 + 1) - M);
        kbv[cls] = kb;
        const int rlo = kFftN - kb - M + 1;                                // ... are entries rlo .. rlo + M - 1 of R (descending bins)
#pragma unroll
        for (int i0 = 0; i0 < kFftN; i0 += kPrepWaves * 64) {
            const int i = i0 + tid;
            const float v = Rs[i];
            if (cls == 0) {
                sums[0] += v * v;
                dcs += i < 64 ? v * v : 0.0f;
                mxdc = fmaxf(mxdc, i < 64 ? v * v : 0.0f);
            }
            const int j = i - rlo;
            const bool in = j >= 0 && j < M;
            sums[1 + 3 * cls] += in ? 0.0f : v * v;
            mxo[cls] = fmaxf(mxo[cls], in ? 0.0f : v * v);
            sums[2 + 3 * cls] += in && j + M / 2 < M ? v * Rs[min(i + M / 2, kFftN - 1)] : 0.0f;
            sums[3 + 3 * cls] += in && j + 3 * M / 4 < M ? v * Rs[min(i + 3 * M / 4, kFftN - 1)] : 0.0f;
            // entry i is bin 2048 - i: above Nyquist for i < 1024, its mirror image (bin i) is entry 2048 - i
            const int jm = kFftN - i - rlo;
            pmir[cls] += in && i < kFftN / 2 && kFftN / 2 - i >= 5 * M / 32 && jm >= 0 && jm < M ? fabsf(v * Rs[(kFftN - i) & (kFftN - 1)]) : 0.0f;
        }
    }
#pragma unroll
    for (int k = 0; k < 7; ++k) {
        const float w = wave_sum(sums[k]);
        if (lane == 0) red[wave][k] = w;
    }
#pragma unroll
    for (int cls = 0; cls < 2; ++cls) {
        const float w = wave_sum(pmir[cls]);
        if (lane == 0) red[wave][10 + cls] = w;
    }
    {
        const float w = wave_sum(dcs);
        float m = mxdc;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
        if (lane == 0) { red[wave][12] = w; red[wave][13] = m; }
    }
#pragma unroll
    for (int cls = 0; cls < 2; ++cls) {
        float m = mxo[cls];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
        if (lane == 0) red[wave][8 + cls] = m;
    }
    __syncthreads();
    if (tid == 0) {
        int flags = 0, need = 0;
#pragma unroll
        for (int cls = 0; cls < 2; ++cls) {
            float v[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int col = k == 0 ? 0 : k + 3 * cls;
                float s = 0.0f;
#pragma unroll
                for (int w = 0; w < kPrepWaves; ++w) s += red[w][col];
                v[k] = s;
            }
            float pm = 0.0f;
#pragma unroll
            for (int w = 0; w < kPrepWaves; ++w) pm += red[w][10 + cls];
            float odc = 0.0f, mdc = 0.0f;
#pragma unroll
            for (int w = 0; w < kPrepWaves; ++w) { odc += red[w][12]; mdc = fmaxf(mdc, red[w][13]); }
            // (the bias-free energy bound counts what is dropped at DC kBandAdjacentDC / kBandAdjacent times too; not under round 5's rule, a.eta = kBandEta)
            const float o2 = a.eta < kBandEta ? v[1] + (kBandAdjacentDC / kBandAdjacent - 1.0f) * odc : v[1];
            bool ok = o2 <= a.eps2 * v[0] && v[2] <= a.eta * v[0] && v[3] <= a.eta * v[0] && pm <= a.eta * v[0];
            // the smallest bias that admits the class beyond the strict rule (band_need): the filter's core is 2 sigma_k around the centre bin
            const int Mc = 256 << cls;
            const float sgc = fminf(fmaxf(kernel[2 * f + 1], bd.sigma_lo), bd.sigma_hi), sk = (float)kFftN / (6.2831853f * sgc);
            const float dmin = (float)min(k0 - (kbv[cls] - 1), kbv[cls] + Mc - k0) - 2.0f * sk;
            const float spw = pool_sigma(pool_w[f], K);
            float mx = 0.0f;
#pragma unroll
            for (int w = 0; w < kPrepWaves; ++w) mx = fmaxf(mx, red[w][8 + cls]);
            int nd = band_need(v[1], mx, v[2], v[3], v[0], a.eta, fabsf(Rs[(kFftN - k0) & (kFftN - 1)]),
                               band_pool_gamma(spw, K, dmin, kFftN), spw, K, kFftN, pm, Mc, odc, mdc);
            if (a.bwd_slabs && !band_deriv_fits(k0, kbv[cls], Mc, sk)) { ok = false; nd = kBandNever; }   // (backward: see kBandDerivCore)
            if (a.force) { ok = a.force == cls + 1; nd = kBandNever; }
            flags |= ok ? 1 << cls : 0;
            need |= nd << (16 * cls);
        }
        if (a.classes) {
            const bool c1 = (flags & 1) || band_bias_admits(a.cls_bias, f, true, need), c2 = (flags & 2) || band_bias_admits(a.cls_bias, f, true, need >> 16);
            a.classes[f] = c1 ? 256 : c2 ? 512 : kFftN;
        }
        a.rec[4 * f] = flags;
        a.rec[4 * f + 1] = kbv[0];
        a.rec[4 * f + 2] = kbv[1];
        a.rec[4 * f + 3] = need;          // bmin(256) | bmin(512) << 16, fp16 codes
    }
}
#endif

// ---- device side of the main kernel -------------------------------------------------------------------------------------
// Every workgroup builds the same plan from the per-filter records (wave 0, before the first barrier): lists of the filters
// per class in filter order; when the 256-point class leaves a partly filled task and its stragglers also pass the 512-point
// criteria, they join the 512-point class if that saves a task.  bl: [0] tasks per block, [1] / [2] 256- / 512-point tasks;
// descriptors (class | index << 2: filter for class 0, first member for the others); members (filter | first bin << 16).
__device__ __forceinline__ void band_build_plan(const int* __restrict__ rec, const int* __restrict__ elist, int n_edge, int F, int* bl, int lane,
                                                const float* __restrict__ bias = nullptr, float smax = 1.0f) {
    // ONE round trip to the records (a 16-byte load per filter) and the edge list; everything after it is LDS and register work
    // (round 5: the plan was four dependent global round trips deep, ~6 k cycles with the other waves at the kernel's first barrier)
    int* tdesc = bl + kBandPlanHead;
    int* mem = tdesc + F + 4;
    int* l0 = mem + F + 16;
    int* l1 = l0 + F;
    int* l2 = l1 + F;
    int* prec = l2 + F;                                                   // per filter: flags | first bin of the 256-point window << 2 | of the 512-point one << 13
    const int ev = lane < 4 * n_edge ? elist[lane] : 0;
    int n0 = 0, n1 = 0, n2 = 0;
    const unsigned long long below = (1ull << lane) - 1ull;
    for (int f0 = 0; f0 < F; f0 += 64) {
        const int f = f0 + lane;
        int4 r4 = make_int4(0, 0, 0, kBandNever | (kBandNever << 16));
        if (f < F) r4 = reinterpret_cast<const int4*>(rec)[f];
        // the classes this call admits: the strict rule, or this call's bias at or above the recorded bmin (band_need above)
        const bool relax = smax > 1.0f;
        const int r = (r4.x & 3) | (f < F && band_bias_admits(bias, f, relax, r4.w) ? 1 : 0) | (f < F && band_bias_admits(bias, f, relax, r4.w >> 16) ? 2 : 0);
        if (f < F) prec[f] = (r & 3) | ((r4.y & 0x7ff) << 2) | ((r4.z & 0x7ff) << 13);
        const bool c1 = f < F && (r & 1), c2 = f < F && !(r & 1) && (r & 2), c0 = f < F && !(r & 3);
        const unsigned long long b0 = __ballot(c0), b1 = __ballot(c1), b2 = __ballot(c2);
        if (c0) l0[n0 + __popcll(b0 & below)] = f;
        if (c1) l1[n1 + __popcll(b1 & below)] = f;
        if (c2) l2[n2 + __popcll(b2 & below)] = f;
        n0 += __popcll(b0);
        n1 += __popcll(b1);
        n2 += __popcll(b2);
    }
    if (lane < 4 * n_edge) bl[4 + lane] = ev;                             // the edge list, for the tasks' edge loops
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
    const int r1 = n1 & 7;
    if (r1) {
        const int f = lane < r1 ? l1[n1 - r1 + lane] : 0;
        const bool ok2 = lane >= r1 || (prec[f] & 2);
        if (__all(ok2) && (n2 + r1 + 3) / 4 <= 1 + (n2 + 3) / 4) {
            if (lane < r1) l2[n2 + lane] = f;
            n2 += r1;
            n1 -= r1;
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
        }
    }
    const int t1 = (n1 + 7) >> 3, t2 = (n2 + 3) >> 2, nt = t1 + t2 + n0;
    for (int t = lane; t < nt; t += 64)
        tdesc[t] = t < t1 ? (1 | ((8 * t) << 2)) : t < t1 + t2 ? (2 | ((8 * t1 + 4 * (t - t1)) << 2)) : (l0[t - t1 - t2] << 2);
    for (int i = lane; i < 8 * t1; i += 64) {
        const int f = l1[min(i, n1 - 1)];
        mem[i] = f | (((prec[f] >> 2) & 0x7ff) << 16) | (i >= n1 ? kBandInvalid : 0);
    }
    for (int i = lane; i < 4 * t2; i += 64) {
        const int f = l2[min(i, n2 - 1)];
        mem[8 * t1 + i] = f | (((prec[f] >> 13) & 0x7ff) << 16) | (i >= n2 ? kBandInvalid : 0);
    }
    if (lane == 0) { bl[0] = nt; bl[1] = t1; bl[2] = t2; }
}
// one stage of the 16-point decimation-in-time transform on registers BASE .. BASE + 15 (position p lives in register
// BASE + brev4(p); twiddles W_16^k = W_32^(2k): the register conventions of fft32_dit_stage)
template <int HALF, int BASE>
__device__ __forceinline__ void band_dit16_stage(float (&re)[32], float (&im)[32]) {
    constexpr float C[16] = {1.0f, 0.98078528f, 0.923879533f, 0.831469612f, 0.707106781f, 0.555570233f, 0.382683432f, 0.195090322f, 0.0f, -0.195090322f, -0.382683432f, -0.555570233f, -0.707106781f, -0.831469612f, -0.923879533f, -0.98078528f};
    constexpr float S[16] = {0.0f, -0.195090322f, -0.382683432f, -0.555570233f, -0.707106781f, -0.831469612f, -0.923879533f, -0.98078528f, -1.0f, -0.98078528f, -0.923879533f, -0.831469612f, -0.707106781f, -0.555570233f, -0.382683432f, -0.195090322f};
#pragma unroll
    for (int blk = 0; blk < 16; blk += 2 * HALF) {
#pragma unroll
        for (int j = 0; j < HALF; ++j) {
            const int a = BASE + brev4(blk + j), b = BASE + brev4(blk + j + HALF);
            constexpr int STEP = 8 / HALF;
            const int tw = 2 * j * STEP;                                    // w = W_16^(j STEP) = W_32^tw
            const float ar = re[a], ai = im[a], br = re[b], bi = im[b];
            if (tw == 0) {
                re[a] = ar + br;
                im[a] = ai + bi;
                re[b] = ar - br;
                im[b] = ai - bi;
            } else if (tw == 8) {                                           // w = -i
                re[a] = ar + bi;
                im[a] = ai - br;
                re[b] = ar - bi;
                im[b] = ai + br;
            } else {
                const float pr = fmaf(bi, -S[tw], fmaf(br, C[tw], ar));
                const float pi = fmaf(bi, C[tw], fmaf(br, S[tw], ai));
                re[a] = pr;
                im[a] = pi;
                re[b] = fmaf(2.0f, ar, -pr);
                im[b] = fmaf(2.0f, ai, -pi);
            }
        }
    }
}

// scr[(16 h + brev4(i)) * 68 + lane] = v[16 h + i]: rows = (half, m2), columns = phase-1 lanes (add-tid stores through M0,
// as wg_transpose_store)
__device__ __forceinline__ void band_transpose_store(const float (&v)[32], unsigned scr_lds) {
    unsigned keep;
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
                 : "v"(v[0]), "v"(v[1]), "v"(v[2]), "v"(v[3]), "v"(v[4]), "v"(v[5]), "v"(v[6]), "v"(v[7]), "v"(v[8]), "v"(v[9]), "v"(v[10]), "v"(v[11]), "v"(v[12]), "v"(v[13]), "v"(v[14]), "v"(v[15]), "s"(scr_lds)
                 : "memory");
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %17\n\ts_nop 0\n\t"
                 "ds_write_addtid_b32 %1 offset:4352\n\t"
                 "ds_write_addtid_b32 %2 offset:6528\n\t"
                 "ds_write_addtid_b32 %3 offset:5440\n\t"
                 "ds_write_addtid_b32 %4 offset:7616\n\t"
                 "ds_write_addtid_b32 %5 offset:4896\n\t"
                 "ds_write_addtid_b32 %6 offset:7072\n\t"
                 "ds_write_addtid_b32 %7 offset:5984\n\t"
                 "ds_write_addtid_b32 %8 offset:8160\n\t"
                 "ds_write_addtid_b32 %9 offset:4624\n\t"
                 "ds_write_addtid_b32 %10 offset:6800\n\t"
                 "ds_write_addtid_b32 %11 offset:5712\n\t"
                 "ds_write_addtid_b32 %12 offset:7888\n\t"
                 "ds_write_addtid_b32 %13 offset:5168\n\t"
                 "ds_write_addtid_b32 %14 offset:7344\n\t"
                 "ds_write_addtid_b32 %15 offset:6256\n\t"
                 "ds_write_addtid_b32 %16 offset:8432\n\t"
                 "s_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(v[16]), "v"(v[17]), "v"(v[18]), "v"(v[19]), "v"(v[20]), "v"(v[21]), "v"(v[22]), "v"(v[23]), "v"(v[24]), "v"(v[25]), "v"(v[26]), "v"(v[27]), "v"(v[28]), "v"(v[29]), "v"(v[30]), "v"(v[31]), "s"(scr_lds)
                 : "memory");
}

// eight ring reads of chunk Q = 2 h + (r >> 3): register (h, r) <- A'[kb + c + (A/2) h + A r]
template <int A, int Q, int... J>
__device__ __forceinline__ void band_rd_chunk(v2f (&av)[8], unsigned a0, std::integer_sequence<int, J...>) {
    (lds_rd8<8 * (A * ((Q & 1) * 8 + J) + (A / 2) * (Q >> 1))>(av[J], a0), ...);
}
// twiddle of register k = 16 h + i: W_M^(j1 m2), m2 = brev4(i), j1 = c + (A/2) h = entry twl[2 m2][(64 / A) j1] of the
// 2048-point table (twl[k1][l] = W_2048^(l k1)); the lane's base is twl + (64 / A) c, (64 / A) (A / 2) h = 32 h
struct OffBandTw {
    static constexpr int off(int k) { return 8 * (2 * 64 * brev4(k & 15) + 32 * (k >> 4)); }
};

// the R values of a band task's bins, register (h, r) <- R[fid][2048 - (kb + c + (A/2) h + A r)] (32 loads, like a spectrum row)
template <int A>
__device__ __forceinline__ void band_load_spectrum(float (&rq)[32], const float* R, int me1, int lane) {
    const int fid = me1 & 0xffff, kb = (me1 >> 16) & 0x7ff, c = lane & (A / 2 - 1);
    const float* src = R + (size_t)fid * kFftN + (kFftN - kb - c);
    asm volatile("" ::: "memory");
#pragma unroll
    for (int k = 0; k < 32; ++k) rq[k] = src[-(A * (k & 15) + (A / 2) * (k >> 4))];
    asm volatile("" ::: "memory");
}

// sum over the lanes of one filter (the lane bits above log2 G), every lane of the filter gets the total
template <int A>
__device__ __forceinline__ float band_filter_sum(float v) {
    if constexpr (A == 32) v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x124, 0xf, 0xf, false));   // row_ror:4
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x128, 0xf, 0xf, false));                          // row_ror:8
    auto a = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    v = __uint_as_float(a[0]) + __uint_as_float(a[1]);
    auto b = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return __uint_as_float(b[0]) + __uint_as_float(b[1]);
}

// Geometry of the blocks a band task runs on.  N4K = false: the 2048-sample plan (A = 16 / 32: 256 / 512 points, D = 8 / 4).
// N4K = true: the 4096-sample plan of the 32 kHz window (leaf_fft_wg4k.hpp): the A = 32 layout at D = 8 -- a 512-bin window of the
// 4096-point spectrum, four filters per task, register rows of 128 samples -- with its own table layout (one class per filter).
__host__ __device__ constexpr int band4k_gz_floats(int K, int hop) { return (band_gz_len_d(K, hop, 32, 8) + 3) / 4 * 4; }
constexpr int kBand4kLS = 3200;                              // valid outputs of a 4096-sample block at K = 801 / hop = 320
template <int A, int SK, int SHOP, bool N4K>
struct BandGeom {
    static constexpr int D = N4K ? 8 : band_d(A);
    static constexpr int LS = N4K ? kBand4kLS : fft_block_len(SK, SHOP, true);
    static constexpr int RL = band_rl(A, D);
    static constexpr int GZF = N4K ? band4k_gz_floats(SK, SHOP) : band_gz_floats(SK, SHOP);   // (constexpr: the gcd in them is not folded otherwise)
    static constexpr int GZ0 = (!N4K && A == 32) ? band_gz_len(SK, SHOP, 16) : 0;
    static constexpr int NCLS = N4K ? 1 : 2, CLS = (!N4K && A == 32) ? 1 : 0;                 // edge tables: classes per filter, this one's index
    static_assert(!N4K || A == 32, "the 4096-sample plan has the 512-point class only");
};

// One band task.  rq: the R values of this task's bins (requested by the previous task); Aring: the block's half spectrum;
// mem: this task's G member entries; mid(): called once the energies are in registers -- it reserves the next task and
// requests ITS 32 table values (exactly 32 loads, so that the wait for this task's pooling weights can be counted; it returns
// the number of loads it issued: 32, or 0 -- the 4096-sample kernel, whose tasks fetch their own table values);
// out(filter, frame, value): the block's share of a frame sum (added to the clip's LDS sums, or stored to the slot the block
// has in the frame ring / the partial-sum buffer); mlo .. mhi: the frames whose window meets the block.
template <int A, int SK, int SHOP, bool N4K = false, typename Mid, typename Out, typename Stamp>
__device__ __forceinline__ void band_task(const FftParams& p, const float (&rq)[32], const float2* Aring, const int* mem, const int* elist,
                                          const float2* twl, float* scr, unsigned scr_lds, int* inv_cnt, int c, int mlo, int mhi, int lane,
                                          Mid&& mid, Out&& out, Stamp&& stamp) {
    using GEO = BandGeom<A, SK, SHOP, N4K>;
    constexpr int LPF = band_lpf(A), D = GEO::D, G = band_d(A), RL = GEO::RL;
    constexpr int PADL = SK / 2 + SK % 2 - 1, LS = GEO::LS;
    constexpr int DMIN = -((SK - 1 - PADL) / SHOP), DMAX = (LS - 1 + PADL) / SHOP, NFR = DMAX - DMIN + 1;
    constexpr int LPHI = kBandLh * D, PG = band_gcd(RL, SHOP), C0MIN = band_c0min_d(SK, SHOP, A, D), NV = band_nv_d(SK, SHOP, A, D);
    static_assert((N4K || band_geometry_ok(SK, SHOP)) && NFR <= 16 && PG % D == 0 && LS % RL == 0 && LS / RL <= 32, "band tasks: static geometry");
    const int me1 = mem[lane / LPF];
    const int kb = (me1 >> 16) & 0x7ff, c1 = lane & (LPF - 1);
    float zre[32], zim[32];
    {
        // Z = conj(A'[k]) R, bins ascending from kb, fused with the first DIT stage of the 16-point transforms over j2:
        // registers (h, r) and (h, r + 8) with unit twiddles (as wg's fused multiply)
        const unsigned a0 = lds_addr(Aring + kb + c1);
        v2f av[4][8];
        constexpr auto seq = std::make_integer_sequence<int, 8>{};
        auto pairs = [&](auto hh) {
            constexpr int h = decltype(hh)::value;
#pragma unroll
            for (int r = 0; r < 8; ++r) {
                const float ra = rq[16 * h + r], rb = rq[16 * h + r + 8];
                const v2f xa = av[2 * h][r], xb = av[2 * h + 1][r];
                const float tr_ = xa.x * ra, ti_ = -(xa.y * ra);
                zre[16 * h + r] = fmaf(xb.x, rb, tr_);
                zim[16 * h + r] = fmaf(-xb.y, rb, ti_);
                zre[16 * h + r + 8] = fmaf(-xb.x, rb, tr_);
                zim[16 * h + r + 8] = fmaf(xb.y, rb, ti_);
            }
        };
        band_rd_chunk<A, 0>(av[0], a0, seq);
        band_rd_chunk<A, 1>(av[1], a0, seq);
        band_rd_chunk<A, 2>(av[2], a0, seq);
        lds_wait8<8>(av[0]);
        lds_wait8<8>(av[1]);
        pairs(std::integral_constant<int, 0>{});
        band_rd_chunk<A, 3>(av[3], a0, seq);
        lds_wait8<0>(av[2]);
        lds_wait8<0>(av[3]);
        pairs(std::integral_constant<int, 1>{});
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::"v"(zre[31]), "v"(zim[31]) : "memory");
    wg_release();
    if (lane == 0) __hip_atomic_fetch_add(inv_cnt, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    stamp(4);                                                            // spectral multiply done
    band_dit16_stage<2, 0>(zre, zim);
    band_dit16_stage<2, 16>(zre, zim);
    band_dit16_stage<4, 0>(zre, zim);
    band_dit16_stage<4, 16>(zre, zim);
    band_dit16_stage<8, 0>(zre, zim);
    band_dit16_stage<8, 16>(zre, zim);                                  // register 16 h + i <-> m2 = brev4(i), column j1 = c1 + (A/2) h
    lds_stream32(lds_addr(twl + (64 / A) * c1), OffBandTw{}, [&](int k, v2f w) {
        if (brev4(k & 15) == 0) return;                                 // W^0 = 1
        const float r = zre[k] * w.x - zim[k] * w.y;
        zim[k] = zre[k] * w.y + zim[k] * w.x;
        zre[k] = r;
    });
    stamp(11);                                                           // first transforms and twiddles done
    // transposition: phase-2 lane = l2 G + g reads the columns of filter g (phase-1 lanes g LPF ..) of its rows
    const int g2 = lane & (G - 1), l2 = lane / G;
    float tr[32], ti[32];
    {
        const f32x4* row = reinterpret_cast<const f32x4*>(scr + l2 * kWgScrStride + g2 * LPF);
        constexpr int R16 = 16 * kWgScrStride / 4, R8 = 8 * kWgScrStride / 4;    // 16 / 8 rows further, in 16-byte units
        auto plane = [&](const float (&src)[32], float (&t)[32]) {
            band_transpose_store(src, scr_lds);
            f32x4 v[8];
            if constexpr (A == 16) {
                v[0] = row[0]; v[1] = row[1]; v[2] = row[R16]; v[3] = row[R16 + 1];                       // m2 = l2: j1 = 0..15
                v[4] = row[R8]; v[5] = row[R8 + 1]; v[6] = row[R8 + R16]; v[7] = row[R8 + R16 + 1];       // m2 = l2 + 8
            } else {
                v[0] = row[0]; v[1] = row[1]; v[2] = row[2]; v[3] = row[3];                               // j1 = 0..15
                v[4] = row[R16]; v[5] = row[R16 + 1]; v[6] = row[R16 + 2]; v[7] = row[R16 + 3];           // j1 = 16..31
            }
#pragma unroll
            for (int q = 0; q < 8; ++q) { t[4 * q] = v[q].x; t[4 * q + 1] = v[q].y; t[4 * q + 2] = v[q].z; t[4 * q + 3] = v[q].w; }
            asm volatile("" ::: "memory");
        };
        pin32(zre);
        pin32(zim);
        plane(zre, tr);
        pin32(tr);
        plane(zim, ti);
        pin32(ti);
    }
    stamp(12);                                                           // transposed
    // the pooling weights of this lane's filter: requested before the second transforms (LEAF_BAND_PW_EARLY) they land under them
    const int me2 = mem[g2];
    const int fid2 = me2 & 0xffff;
    const bool valid = !(me2 & kBandInvalid);
    float pw[NV];
    auto load_weights = [&]() {
        const float* gsrc = p.band.gz + (size_t)fid2 * GEO::GZF + GEO::GZ0 + l2;
        asm volatile("" ::: "memory");
#pragma unroll
        for (int k = 0; k < NV; ++k) pw[k] = gsrc[PG / D * k];
        asm volatile("" ::: "memory");
    };
    if constexpr (LEAF_BAND_PW_EARLY) load_weights();
    if constexpr (A == 16) {
        band_dit16_stage<1, 0>(tr, ti);
        band_dit16_stage<1, 16>(tr, ti);
        band_dit16_stage<2, 0>(tr, ti);
        band_dit16_stage<2, 16>(tr, ti);
        band_dit16_stage<4, 0>(tr, ti);
        band_dit16_stage<4, 16>(tr, ti);
        band_dit16_stage<8, 0>(tr, ti);
        band_dit16_stage<8, 16>(tr, ti);                                 // register 16 h + i <-> m1 = brev4(i), m2 = l2 + 8 h
    } else {
        fft32_dif(tr, ti);                                               // register i <-> m1 = brev5(i), m2 = l2
    }
    stamp(5);                                                            // transforms done
    if constexpr (!LEAF_BAND_PW_EARLY) load_weights();
    float e[32];
#pragma unroll
    for (int k = 0; k < 32; ++k) e[k] = tr[k] * tr[k] + ti[k] * ti[k];
    pin32(e);                                                            // every energy is in its register (tr / ti are dead) before ...
    if (mid()) asm volatile("s_waitcnt vmcnt(32)" ::: "memory");          // next task reserved, its 32 table values requested: the pooling weights
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                 // (issued before those 32 loads) have landed
    stamp(6);                                                            // energies, next task reserved, weights landed
    // Edge frames of this block (block 0 and the clip's last blocks only): dense tables over all 32 registers (the tails wrap
    // around the block).  The first entry's table is requested before the reduction of the regular frames, each further one before the
    // previous is consumed (those unconditionally -- a clamped entry when there is none -- so that the waits can be counted).
    const int n_edge = p.band.n_edge;
    auto next_edge = [&](int s) {
        while (s < n_edge && __builtin_amdgcn_readfirstlane(elist[4 * s]) != c) ++s;
        return s;
    };
    const bool has_edges = c == 0 || c >= p.nblk - 2;                     // a clip's interior blocks have none
    const float* etab = p.band.edge + ((size_t)fid2 * GEO::NCLS + GEO::CLS) * kBandMaxEdge * 512 + l2;
    auto issue_edge = [&](float (&et)[32], int s) {
        const float* tab = etab + (size_t)min(s, kBandMaxEdge - 1) * 512;
        asm volatile("" ::: "memory");
#pragma unroll
        for (int k = 0; k < 32; ++k) et[k] = tab[k * LPF];
        asm volatile("" ::: "memory");
    };
    float et0[32], et1[32];
    int s0 = n_edge;
    float acc[16];
#pragma unroll
    for (int fi = 0; fi < 16; ++fi) acc[fi] = 0.0f;
#pragma unroll
    for (int rho = 0; rho < LS / RL; ++rho) {
        const int k = A == 32 ? brev5(rho) : 16 * (rho & 1) + brev4(rho >> 1);   // register of row rho
#pragma unroll
        for (int fi = 0; fi < NFR; ++fi) {
            const int c0 = RL * rho - ((DMIN + fi) * SHOP - PADL);
            if (c0 >= C0MIN && c0 <= SK - 1 + LPHI) acc[fi] = fmaf(e[k], pw[(c0 - C0MIN) / PG], acc[fi]);
        }
    }
    asm volatile("" : "+v"(acc[0]));
    if (LEAF_BAND_EDGE_EARLY && has_edges) {                             // (the pooling weights are dead: registers for the first table)
        s0 = next_edge(0);
        if (s0 < n_edge) issue_edge(et0, s0);
    }
    // halving butterfly over the lanes of a filter (frame_butterfly16 without the stages inside a filter's lanes):
    // afterwards acc[0] (and acc[1] for A = 16) hold the totals of frame fi0 (+ 1)
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        auto g = __builtin_amdgcn_permlane32_swap(__float_as_uint(acc[i]), __float_as_uint(acc[i + 8]), false, false);
        acc[i] = __uint_as_float(g[0]) + __uint_as_float(g[1]);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        auto g = __builtin_amdgcn_permlane16_swap(__float_as_uint(acc[i]), __float_as_uint(acc[i + 4]), false, false);
        acc[i] = __uint_as_float(g[0]) + __uint_as_float(g[1]);
    }
    const bool up8 = (lane & 8) != 0, up4 = (lane & 4) != 0;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const float send = up8 ? acc[i] : acc[i + 2], keep = up8 ? acc[i + 2] : acc[i];
        acc[i] = keep + __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(send), 0x128, 0xf, 0xf, false));   // row_ror:8
    }
    if constexpr (A == 32) {
        const float send = up4 ? acc[0] : acc[1], keep = up4 ? acc[1] : acc[0];
        int t = __builtin_amdgcn_update_dpp(0, __float_as_int(send), 0x104, 0xf, 0x5, false);   // row_shl:4 -> banks 0, 2
        t = __builtin_amdgcn_update_dpp(t, __float_as_int(send), 0x114, 0xf, 0xa, false);       // row_shr:4 -> banks 1, 3
        acc[0] = keep + __int_as_float(t);
    }
    const int n_c = c * LS;
    {
        // regular frames the block meets (the widened window meets the same blocks as the window itself: band_geometry_ok)
        const int rlo = max(mlo, p.band.reg_lo), rhi = min(mhi, p.band.reg_hi);
        const int b5 = (lane >> 5) & 1, b4 = (lane >> 4) & 1, b3 = (lane >> 3) & 1, b2 = (lane >> 2) & 1;
        constexpr int NOUT = A == 16 ? 2 : 1;
#pragma unroll
        for (int o = 0; o < NOUT; ++o) {
            const int fi = A == 16 ? 8 * b5 + 4 * b4 + 2 * b3 + o : 8 * b5 + 4 * b4 + 2 * b3 + b2;
            const int m = n_c / SHOP + DMIN + fi;
            if (valid && fi < NFR && m >= rlo && m <= rhi) out(fid2, m, acc[o]);
        }
    }
    stamp(13);                                                           // pooling, reduction, sums added
    if (!LEAF_BAND_EDGE_EARLY && has_edges) {
        s0 = next_edge(0);
        if (s0 < n_edge) issue_edge(et0, s0);
    }
    auto consume_edge = [&](const float (&et)[32], int s) {
        float v = 0.0f;
#pragma unroll
        for (int k = 0; k < 32; ++k) v = fmaf(e[k], et[k], v);
        v = band_filter_sum<A>(v);
        if (valid && l2 == 0) out(fid2, elist[4 * s + 1], v);
    };
#if LEAF_BAND_EDGE_PIPE
    while (s0 < n_edge) {
        const int s1 = next_edge(s0 + 1);
        issue_edge(et1, s1);
        consume_edge(et0, s0);
        if (s1 >= n_edge) break;
        s0 = next_edge(s1 + 1);
        issue_edge(et0, s0);
        consume_edge(et1, s1);
    }
#else
    (void)et1;
    while (s0 < n_edge) {
        consume_edge(et0, s0);
        s0 = next_edge(s0 + 1);
        if (s0 < n_edge) issue_edge(et0, s0);
    }
#endif
}

}  // namespace
