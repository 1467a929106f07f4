/*
 * leaf_hip.h -- C ABI of the MI355X-native LEAF frontend (libleaf_hip.so, gfx950 only).
 *
 * This is the drop-in boundary for the hot path of SarthakYadav/leaf-pytorch:
 *     leaf_pytorch/frontend.py:78-89   Leaf.forward
 * The reference has no native code and therefore no FFI; each entry point below names the
 * reference function(s) (file:line under the reference repo) whose arithmetic it replaces.
 * The Python host side (leaf_pytorch_amd/frontend.py) binds these with ctypes; PyTorch is only
 * used there for device memory and the current HIP stream.  See INTEGRATION.md.
 *
 * Conventions
 *   - plain pointers + sizes, no torch types; every pointer is DEVICE memory (HBM), fp32,
 *     contiguous, owned by the caller; the library never allocates, frees or retains pointers.
 *   - `stream` is a hipStream_t passed as void* (NULL = the default stream).  All work is enqueued
 *     asynchronously on that stream; nothing synchronises the host.
 *   - re-entrant; nothing is kept between calls except two process-wide, write-once caches: the CU count per
 *     device ordinal, and ONE environment switch read at first use -- LEAF_NO_4K=1 keeps every window on the
 *     2048-sample plan (a test / A-B switch for the 4096-sample kernels; it changes which kernel runs, never the
 *     results beyond fp32 rounding).  Per-call options travel in `algo` / `flags`, never in setters.  Parameters are
 *     re-read on every call (they are learnable: the clamps of the reference are applied functionally, never
 *     written back).
 *   - a WORKSPACE belongs to one call at a time: calls that may execute concurrently (different streams) need workspaces of
 *     their own -- kernels of one call hand data to each other through it (partial sums, tables, the seam slots of the
 *     one-launch kernel's two halves), and a second call writing the same bytes would be read as the first call's.  Calls on ONE
 *     stream may share a workspace (they execute in order).
 *   - return value: LEAF_OK (0) or a negative leaf_status code.  Never throws, never aborts.
 *   - B = 0 is the EMPTY BATCH, not an error (the reference returns a (0, F, T') tensor: frontend.py:78-89 ->
 *     convolution.py:97): leaf_forward_f32 / _save_f32 / _prepared_f32 / _profiled_f32 return LEAF_OK without a launch
 *     (x / out / workspace may be NULL), leaf_backward_f32 zero-fills the parameter gradients (the sum over no clips)
 *     and touches nothing else; the workspace queries return 0.  T, F, K, hop must still be >= 1.
 *
 * Shapes (reference notation):
 *   B batch, T samples per clip, F = n_filters, K = window size in samples, hop = stride in samples,
 *   T' = leaf_num_frames(T, K, hop) = floor((T + padL + padR - K) / hop) + 1 (= floor((T-1)/hop)+1).
 */
#ifndef LEAF_HIP_H_
#define LEAF_HIP_H_

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define LEAF_ABI_VERSION 5

typedef enum leaf_status {
    LEAF_OK = 0,
    LEAF_ERR_NULL_POINTER = -1,   /* a required pointer argument is NULL                    */
    LEAF_ERR_BAD_SHAPE = -2,      /* T,F,K,hop out of range (all >= 1), B < 0 or B*T >= 2^31 */
    LEAF_ERR_WORKSPACE = -3,      /* workspace missing or smaller than leaf_workspace_bytes */
    LEAF_ERR_BAD_ALGO = -4,       /* unknown / inapplicable algorithm selector              */
    LEAF_ERR_LAUNCH = -5,         /* HIP reported a launch failure (hipGetLastError != 0)   */
    LEAF_ERR_NO_DEVICE = -6,      /* no usable gfx950 device                                */
    LEAF_ERR_ALIGNMENT = -7,      /* a buffer is not 4-byte aligned                         */
    LEAF_ERR_UNSUPPORTED = -8     /* valid arguments, unsupported combination: bfloat16 I/O with a backward
                                     (leaf_forward_save_f32) or with the staged kernels; LEAF_FLAG_PEAKNORM with
                                     leaf_forward_save_f32 / leaf_forward_prepared_f32 or off the overlap-save paths */
} leaf_status;

/* flags */
#define LEAF_FLAG_PCEN   0x1   /* apply PCEN (requires alpha, delta, root, ema_w)                */
#define LEAF_FLAG_LOG1P  0x2   /* extension (not in the reference): out = log1p(pooled), PCEN off */
#define LEAF_FLAG_BWD_STAGED 0x8 /* leaf_backward_f32 only: force the staged (one-lane-per-output) kernels */
#define LEAF_FLAG_BWD_MFMA 0x10 /* leaf_backward_f32 only: force the fused MFMA backward (skip the overlap-save FFT one) */
#define LEAF_FLAG_BWD_FULL_TRANSFORMS 0x40 /* leaf_backward_f32 only (ABI 4): no band-limited filter tasks in the backward -- by default the
                                  static 16 kHz and 32 kHz backwards (K = 401 / hop = 160: parameter gradients from 3/8 block per CU,
                                  with dL/dx at every batch; K = 801 / hop = 320 on 4096-sample blocks: parameter gradients only) give
                                  the filters the forward runs on short transforms their gradients at the decimated rate too
                                  (leaf_band_bwd.hpp): within ~1e-5 of the full-transform gradients' largest component */
#define LEAF_FLAG_BWD_STRICT_BAND_CLASSES 0x80 /* leaf_backward_f32 only (ABI 5): the backward's band tasks decide their classes by round 5's rule
                                  alone; by default they take the forward's decision, which follows the pooling bias of the call
                                  (LEAF_ALGO_STRICT_BAND_CLASSES below) -- measured: gradients stay at ~1e-6 of their column's largest.
                                  Either way a backward launch adds its own condition (leaf_band.hpp band_deriv_fits): the window
                                  holds the DERIVATIVE spectra d/dmu, d/dsigma of the filter, which are wider than the filter; a
                                  filter that fails it takes the next wider class in the backward only */
#define LEAF_FLAG_PEAKNORM 0x20 /* forward only, overlap-save paths (LEAF_ALGO_AUTO / _FFT / _FFT_WG where their plan fits; else
                                  LEAF_ERR_UNSUPPORTED): the result is that of the forward applied to the PEAK-NORMALISED clips
                                  (utilities/data/raw_transforms.py:334-345, the last transform of every reference data
                                  pipeline: a clip whose peak |x| exceeds 1 is divided by that peak) without the normalised
                                  waveform ever being written: one read-only pre-pass finds each clip's scale s, and because the
                                  path is linear up to |.|^2, s^2 multiplies the pooled energies where the bias is added
                                  (pooling.py:41).  Equal to leaf_peak_normalize_f32 + forward up to fp32 rounding (~1e-7). */
#define LEAF_FLAG_IO_BF16 0x4  /* extension (BASELINE configs[4]): x and out are bfloat16 buffers (2 bytes per element),
                                  arithmetic stays fp32; fused path only */

/* algorithm selector for the fused path */
#define LEAF_ALGO_AUTO   0     /* _FFT_SMALL for a handful of clips of a LEAF geometry; else the FFT kernels when their plan fits and K >= 224 or the geometry has a static instance, else MFMA, else staged */
                               /* NOTE: the algorithms agree to ~1e-6 relative, not bit for bit, so under AUTO a clip's output bits depend on
                                  which kernel its batch lands on: they change at the batch thresholds (B * F <= 2 #CUs: _FFT_SMALL; from
                                  ~7/16 block per CU: _FFT_WG; below: _FFT), with the device's CU count and with
                                  LEAF_ALGO_RESERVE_CUS.  Within ONE algorithm a clip's bits do not depend on the batch: pass an explicit
                                  selector where batch-invariant bits matter (tests/test_gpu_dropin.py pins both behaviours). */
#define LEAF_ALGO_STAGED 1     /* unfused stage kernels (materialises every intermediate)    */
#define LEAF_ALGO_MFMA   2     /* fused symmetric-Gabor fp32-MFMA kernel + finalize kernel   */
#define LEAF_ALGO_FFT    3     /* fused overlap-save FFT kernel (2048-point, one wave per block) + finalize kernel */
#define LEAF_ALGO_FFT_WG 4     /* overlap-save, one workgroup per block: the block's spectrum computed once and shared
                                  through LDS by 9-16 waves; static instances for the 16 / 32 / 8 kHz LEAF geometries,
                                  run-time geometry for every other window the plans cover (2048-sample blocks: 64..1216
                                  taps, odd or even; 4096-sample blocks: K = 801 / hop 320 and odd windows 833..2049);
                                  what AUTO picks from about half a block per CU.  2048-sample plan: same tables,
                                  workspace and finalize kernel as LEAF_ALGO_FFT. */

#define LEAF_ALGO_FFT_SMALL 5  /* a handful of clips (test.py:57-71: inference on 1 s chunks) in ONE launch: one workgroup per
                                  (clip, filter) builds the filter's spectrum and pooling weights itself, transforms the
                                  clip's blocks, pools, and runs bias / floor / EMA / PCEN of its row -- no table kernel, no
                                  partial sums in HBM, no row kernel.  16 kHz and 8 kHz LEAF geometries (401/160, 201/80),
                                  B * F <= 2 #CUs (two rounds of workgroups since round 5: 7 .. 12 clips of the default front end 41 -> 30 us), clips of up to 20 blocks; what AUTO picks there.  While 2 B F <= #CUs (clips
                                  of 2..10 blocks) TWO workgroups of seven waves serve a (clip, filter) -- each a half of the
                                  row's frames, the EMA state at the seam handed over through the workspace under a 64-bit
                                  per-launch ticket (ABI 4); a clip's bits are the same in both forms.  The EMA recurrence
                                  runs as a lane scan here: the smoothed value agrees with the other algorithms' sequential
                                  loop to ~1e-7 relative, not to the bit.  Workspace: the per-clip scales of
                                  LEAF_FLAG_PEAKNORM + 16 bytes per (clip, filter) for the seam (leaf_workspace_bytes). */

/* tuning override (tools/ only), OR-ed into `algo`: the fused kernel delays the second wave of every SIMD by
 * n * s_sleep(127) once at start; without it the delay is derived from the geometry. */
#define LEAF_ALGO_TUNE_DESYNC(n) (((n) + 1) << 8)

/* CU reservation, OR-ed into `algo` (forward entry points and leaf_workspace_bytes): this call sizes its persistent kernels
 * for (#CUs - k) compute units, 0 <= k <= 255, leaving k CUs free for kernels of OTHER streams that must make progress
 * while it runs -- the RCCL kernel of the feature all-gather on a side stream: the default kernels keep one
 * workgroup with ~all of a CU's LDS resident on every CU for the whole launch, so a collective launched beside them would
 * otherwise only be scheduled when a launch retires.  Per call, no setter, no state kept between calls. */
#define LEAF_ALGO_RESERVE_CUS(k) (((k) & 0xff) << 16)

/* Streaming finalize, OR-ed into `algo` (forward entry points; static LEAF geometries on the workgroup kernel, batches that
 * give every workgroup whole clips -- otherwise ignored): the per-frame partial sums stay in an LDS *ring* and each block's
 * completed frames are finalized (bias, floor, EMA, PCEN) by the wave that finishes the block's last filter; the finished
 * values are staged in LDS and leave in 128-byte row segments.  No partial-sum buffer in HBM, no second kernel, no tail.
 * Same bits as the default path.  Without the flag it runs exactly where the alternative would be a round trip of the
 * partial sums through HBM: whole clips per workgroup whose frame sums do not fit the LDS (several clips per workgroup,
 * long clips).  Where they do fit (one 1 s clip per workgroup) the default keeps them in LDS and finalizes in the kernel's
 * tail, measured ~1 % faster; the flag selects the streaming form there too. */
#define LEAF_ALGO_STREAM_FINALIZE (1 << 25)

/* Full transforms, OR-ed into `algo` (forward entry points): switches the band-limited filter tasks off.  By default the
 * static 16 kHz workgroup kernel (K = 401, hop = 160) runs every filter whose spectrum -- decided per call on the device from
 * the table the call has just built, i.e. from the CURRENT clamped (mu, sigma) -- holds all but 9e-12 of its energy inside 256
 * or 512 of the 2048 bins on a 256- / 512-point inverse transform of those bins, eight / four filters per task, and pools
 * |y|^2 at the decimated rate (leaf_band.hpp); the static 32 kHz kernel (K = 801, hop = 320, 4096-sample
 * blocks; ABI 4) likewise with one class: a 512-bin window of the 4096-point spectrum, four filters per task, decimation 8.
 * The result differs from the full-transform path by <= ~1e-6 relative (north star: 1e-4); with this flag the
 * call runs the 2048- / 4096-point task for every filter, as before round 5.  leaf_forward_save_f32 (the training forward) takes the
 * band tasks as well (the saved pooled tensor differs by ~1e-6; leaf_backward_f32 has its own band tasks and its own switch,
 * LEAF_FLAG_BWD_FULL_TRANSFORMS), and so does
 * leaf_forward_prepared_f32 when its workspace is sized as documented. */
#define LEAF_ALGO_FULL_TRANSFORMS (1 << 26)

/* Strict band classes, OR-ed into `algo` (forward entry points; ABI 5).  Since round 6 the class decision above also admits a
 * filter whose window drops MORE than 9e-12 of its energy where the pooling BIAS of this call makes that harmless
 * (pooling.py:21-22,31-42): a pooled value is p = bias_f + sum g |y|^2 >= bias_f, and for |x| <= 1 what a window drops adds at most
 * G_0 max_{k outside} R_k^2 / 2 to it (one full-scale tone on the largest dropped bin), so the class is taken when
 * bias_f >= 6 G_0 max R_k^2 / (2 * 5e-6) -- at most 5e-6 of any output; the 6 covers a window edge at DC / Nyquist, where kept
 * components beat against their own dropped images -- provided the filter's pooling window (pool_w) low-passes the cross term
 * between the filter's core and the dropped part: to round 5's level, 6e-6 of a frame's energy, for equal amplitudes, and to 5e-6
 * of the output for a weak kept component next to a strong dropped one, which again asks for a minimal bias (leaf_band.hpp:
 * band_need; one-sample pooling windows take the bias-free part of the rule).  At the default bias 1.0 and pooling width 0.4 this
 * admits four more of the 40 default 16 kHz filters (sigma = 48 samples) and 23 more of the 80 default 32 kHz ones (sigma = 96) to the
 * band tasks.  Two more changes of round 6 ride on the same switch.  (a) On the 2048-sample plan the forward's windows may cross
 * Nyquist (the kernel keeps bins 0..1151 of a block's spectrum; a real block's bins above 1024 mirror those below): a filter whose pass
 * band reaches beyond pi -- the top two of the default 16 kHz bank, anything trained against the clamp of convolution.py:15-22 -- is
 * centred in its window instead of cut by one that ends at Nyquist.  (b) The aliasing bound: two spectral lines more than ~0.3 M bins
 * apart inside an M-bin window beat where the decimated grid cannot represent them; round 5's bound (2e-4 of the filter's energy at lag
 * M / 2) let sigma = 15 - 16 samples onto 256 points, where two tones of amplitude 0.5 at +- 60 bins of the centre were off by 1.5e-4 of
 * (bias 0.1 + pooled energy) on a clip's first frame (profiles/r06/band_alias_pairs.txt).  The pair sums may now reach 1e-5 of the filter's
 * energy only under a minimal bias derived from them (mirror-image pairs of a window across Nyquist included; leaf_band.hpp band_need) and
 * 1e-6 without one; a bias <= 6e-5 (or NaN) takes the bias-free part of the rule.
 * The tables do not depend on the bias (the prep kernels record the smallest admissible bias per filter and class); the decision is
 * taken by the forward kernel from the pool_b of the call.  With this flag round 5's rule applies -- its energy and aliasing bounds,
 * windows inside the half spectrum: its decision, bit for bit.  The backward's band tasks take the same decision, with windows inside
 * the half spectrum (their own switch: LEAF_FLAG_BWD_STRICT_BAND_CLASSES). */
#define LEAF_ALGO_STRICT_BAND_CLASSES (1 << 27)

int leaf_abi_version(void);
const char* leaf_status_string(int status);

/* utils.py:5-10 (padding) + the strided-conv output length used by pooling.py:41. */
int leaf_num_frames(int T, int K, int hop);

/* Bytes of device scratch leaf_forward_f32 / leaf_pool_f32 need for this problem and algo. */
size_t leaf_workspace_bytes(int B, int T, int F, int K, int hop, int algo);

/*
 * Whole forward: frontend.py:78-89 (GaborConv1d -> SquaredModulus -> GaussianLowPass -> max(.,1e-5)
 * -> PCENLayer).
 *   x        [B][T]        waveform (the reference's (B,1,T) with the unit channel dropped)
 *   kernel   [F][2]        _complex_conv._kernel (mu, sigma), unclamped   (convolution.py:58)
 *   pool_w   [F]           _pooling.weights (1,1,F,1) flattened, unclamped (pooling.py:18-20)
 *   pool_b   [F]           _pooling._bias                                  (pooling.py:21-22)
 *   alpha, delta, root, ema_w [F]  _compression.{alpha,delta,root,ema._weights}
 *                          (postprocessing.py:52-54,11); ignored (may be NULL) without LEAF_FLAG_PCEN
 *   out      [B][F][T']
 */
int leaf_forward_f32(const float* x, int B, int T,
                     const float* kernel, const float* pool_w, const float* pool_b,
                     const float* alpha, const float* delta, const float* root, const float* ema_w,
                     int F, int K, int hop, int flags, int algo,
                     float* out, void* workspace, size_t workspace_bytes, void* stream);

/*
 * Training forward: leaf_forward_f32 that additionally stores pooled_raw [B][F][T'] = bias + pooled energy BEFORE
 * the 1e-5 floor (pooling.py:41 output).  Passing it to leaf_backward_f32 saves the backward one filterbank pass.
 */
int leaf_forward_save_f32(const float* x, int B, int T,
                          const float* kernel, const float* pool_w, const float* pool_b,
                          const float* alpha, const float* delta, const float* root, const float* ema_w,
                          int F, int K, int hop, int flags, int algo,
                          float* out, float* pooled_raw, void* workspace, size_t workspace_bytes, void* stream);

/*
 * Measurement variant of leaf_forward_f32 (algo = LEAF_ALGO_AUTO, _MFMA or _FFT): same work on `stream`, bracketed by
 * HIP events recorded on that stream.  Blocks the host until the forward has finished and returns in
 * stage_ms[0..2] the device time (ms) of {table/spectrum preparation, fused filterbank+pool kernel(s), finalize/PCEN
 * kernel}.  Used by bench.py for the per-kernel roofline; not for production calls.
 */
int leaf_forward_profiled_f32(const float* x, int B, int T,
                              const float* kernel, const float* pool_w, const float* pool_b,
                              const float* alpha, const float* delta, const float* root, const float* ema_w,
                              int F, int K, int hop, int flags, int algo,
                              float* out, void* workspace, size_t workspace_bytes, void* stream,
                              float* stage_ms /* host, 3 floats */);
/* which algorithm LEAF_ALGO_AUTO resolves to for this problem (LEAF_ALGO_FFT_SMALL / _FFT_WG / _FFT / _MFMA / _STAGED) */
int leaf_auto_algo(int B, int T, int F, int K, int hop);
/* Plan of the overlap-save path for this problem (measurement / roofline arithmetic in bench.py; no reference
 * counterpart): info[0..7] (host ints) = {transform length N, valid outputs per block L, blocks per clip, filters per
 * task, filter groups, partial-sum slots per frame, pooling-row LDS buffers, dynamic LDS bytes per workgroup} -- of the
 * 4096-sample plan (N = 4096) when that is what LEAF_ALGO_AUTO runs for this problem, else of the 2048-sample plan.
 * LEAF_ERR_BAD_ALGO when neither covers the geometry. */
int leaf_fft_plan_info(int B, int T, int F, int K, int hop, int* info);
/* Which inverse-transform length each filter gets from the band-limited filter tasks (LEAF_ALGO_FULL_TRANSFORMS above;
 * leaf_band.hpp) for the CURRENT parameters (measurement / roofline arithmetic in bench.py, tests; no reference counterpart):
 * classes[f] (DEVICE int32 [F]) = 256, 512 or 2048 -- the decision the forward makes on the device from the filter's own
 * spectrum (convolution.py:15-22 clamps, impulse_responses.py:5-16 taps): all but 9e-12 of the energy of R_f inside the
 * window, and the autocorrelation of |R_f| at lags M/2 and 3M/4 below 2e-4 of its energy.  The forward may still run a
 * 256-class filter on 512 points to fill a task.  On the 4096-sample plan (K = 801, hop = 320) there is one class: 512 (a
 * 512-bin window of the 4096-point spectrum, four filters per task) or 4096.  workspace >= max(leaf_fft_tables_bytes(F, K, hop),
 * leaf_workspace_bytes(1, 8192, F, K, hop, LEAF_ALGO_FFT_WG)).  LEAF_ERR_UNSUPPORTED for a geometry without band tasks (every
 * filter on 2048- / 4096-point transforms).  pool_b (ABI 5; DEVICE [F], may be NULL): the pooling biases the decision is taken
 * for -- what a forward call with these biases runs (LEAF_ALGO_STRICT_BAND_CLASSES above: round 6's bounds, windows that may cross
 * Nyquist on the 2048-sample plan); NULL: round 5's decision, the one quoted in this comment (what the flag runs). */
int leaf_band_classes_f32(const float* kernel, const float* pool_w, const float* pool_b, int F, int K, int hop, int* classes,
                          void* workspace, size_t workspace_bytes, void* stream);

/*
 * Backward of the whole forward (what autograd derives for frontend.py:78-89): given grad_out = dL/d out
 * [B][F][T'], writes dL/d parameter for the seven parameters (same shapes as the inputs; g_alpha..g_ema_w are
 * ignored without LEAF_FLAG_PCEN) and, when g_x != NULL, dL/d x [B][T].  Clamp sub-gradients follow
 * torch.clamp / torch.min / torch.max / torch.maximum as used by the reference (convolution.py:19-20,
 * impulse_responses.py:75, postprocessing.py:14,63-64, frontend.py:84).  Every forward intermediate is recomputed on
 * the device.  Default for windows of 224 .. 1216 taps, odd or even (incl. the reference's default 401/160), and odd windows
 * up to 2049 taps: overlap-save backward (the forward FFT kernels with a backward epilogue: transposed pooling, a second
 * transform, and the tap gradient as two spectral dot products per block and filter; 4096-sample blocks for the 32 kHz
 * geometry and for windows from 833 taps).  With g_x != NULL the same kernels also yield dL/dx for every window up to
 * 1216 taps (the block's spectral gradient summed over its filters, one more transform per block; deterministic, no
 * atomics; on 4096-sample blocks at the 32 kHz geometry K = 801 / hop 320, on 2048-sample blocks elsewhere).  Otherwise, or with LEAF_FLAG_BWD_MFMA: fused MFMA path (filterbank recompute with a backward epilogue that
 * writes dL/dy time-major, then the tap-gradient GEMM dH = S^T dY on the MFMA).  With g_x != NULL beyond 1216 taps,
 * LEAF_FLAG_BWD_STAGED or a geometry neither covers: staged one-lane-per-output kernels.  Workspace =
 * leaf_backward_workspace_bytes for the SAME flags and need_dx = (g_x != NULL): sized for the path that will actually
 * run (a few MB for the overlap-save backward; the staged path materialises dL/dy, B*T*2F floats).
 */
size_t leaf_backward_workspace_bytes(int B, int T, int F, int K, int hop, int flags, int need_dx);
int leaf_backward_f32(const float* x, int B, int T,
                      const float* kernel, const float* pool_w, const float* pool_b,
                      const float* alpha, const float* delta, const float* root, const float* ema_w,
                      int F, int K, int hop, int flags, const float* grad_out,
                      const float* pooled_raw /* from leaf_forward_save_f32, or NULL = recompute */,
                      float* g_kernel, float* g_pool_w, float* g_pool_b,
                      float* g_alpha, float* g_delta, float* g_root, float* g_ema_w,
                      float* g_x, void* workspace, size_t workspace_bytes, void* stream);

/*
 * Stage entry points (each is one reference module's forward; they are what the sub-modules of
 * leaf_pytorch_amd.Leaf call when used on their own, and what the parity tests probe).
 */

/* convolution.py:15-22 + impulse_responses.py:5-16,19-63,66-71: constrained Gabor taps,
 * taps [2F][K], row 2f = Re h_f, row 2f+1 = Im h_f (the layout convolution.py:88-90 feeds conv1d). */
int leaf_gabor_taps_f32(const float* kernel, int F, int K, float* taps, void* stream);

/* impulse_responses.py:74-80: un-normalised Gaussian pooling windows, window [F][K]. */
int leaf_lowpass_window_f32(const float* pool_w, int F, int K, float* window, void* stream);

/* convolution.py:71-99 (GaborConv1d.forward): y [B][2F][T], zero "same" padding, stride 1, no bias.
 * workspace must hold 2*F*K floats. */
int leaf_gabor_conv_f32(const float* x, int B, int T, const float* kernel, int F, int K,
                        float* y, void* workspace, size_t workspace_bytes, void* stream);

/* frontend.py:15-19 (SquaredModulus.forward): y [B][2F][T] -> e [B][F][T] = re^2 + im^2. */
int leaf_squared_modulus_f32(const float* y, int B, int F, int T, float* e, void* stream);

/* pooling.py:31-42 (GaussianLowPass.forward): e [B][F][T] -> pooled [B][F][T'] (+bias, no floor).
 * pool_b may be NULL (use_bias=False).  workspace must hold F*K floats. */
int leaf_gaussian_lowpass_f32(const float* e, int B, int F, int T, const float* pool_w, const float* pool_b,
                              int K, int hop, float* pooled, void* workspace, size_t workspace_bytes,
                              void* stream);

/* postprocessing.py:13-28 (ExponentialMovingAverage.forward): p [B][F][T'] -> ema [B][F][T']. */
int leaf_ema_f32(const float* p, int B, int F, int TP, const float* ema_w, float* ema, void* stream);

/* postprocessing.py:62-69 (PCENLayer.forward) with floor = `floor_` (frontend.py:70 passes 1e-12):
 * p [B][F][T'] -> out [B][F][T']. */
int leaf_pcen_f32(const float* p, int B, int F, int TP, const float* alpha, const float* delta,
                  const float* root, const float* ema_w, float floor_, float* out, void* stream);

/* postprocessing.py:62-69 with the smoother's state handed in and out, for CHUNKED real-time use of the frontend (SURVEY 8f:
 * "very-long-clip time tiling with EMA carry"): p [B][F][n] are the floored pooled frames of one chunk of a running stream,
 * ema_in [B][F] the smoother state after the previous chunk (NULL at the start of a stream: the state starts at the first
 * frame, postprocessing.py:15), ema_out [B][F] receives the state after this chunk (may alias ema_in).  The arithmetic is
 * the fused paths' (the reference's recurrence in its fp32 operation order; the cancellation-free PCEN form), so a stream
 * fed in chunks equals the same frames finalized in one call.  alpha == NULL: no PCEN, out = log1p(p) if `log1p_` else p
 * (then ema_in / ema_out / the PCEN parameters are ignored). */
int leaf_pcen_stream_f32(const float* p, int B, int F, int n, const float* alpha, const float* delta, const float* root,
                         const float* ema_w, float floor_, int log1p_, const float* ema_in, float* ema_out, float* out,
                         void* stream);

/*
 * Stage backwards: the gradient autograd derives for each of the modules above when it is called ON ITS OWN (the
 * reference's sub-modules are ordinary differentiable nn.Modules; Leaf.forward as a whole has leaf_backward_f32).
 * One-lane-per-output kernels, every intermediate materialised; clamp sub-gradients as torch.clamp gives them.
 * Nullable outputs are skipped.  Workspace: leaf_stage_backward_workspace_bytes(stage, B, T, F, K, hop) -- for the EMA
 * and PCEN stages pass T = T' (frames) and K = hop = 1.
 */
#define LEAF_STAGE_GABOR_CONV 1
#define LEAF_STAGE_LOWPASS    2
#define LEAF_STAGE_EMA        3
#define LEAF_STAGE_PCEN       4
size_t leaf_stage_backward_workspace_bytes(int stage, int B, int T, int F, int K, int hop);

/* convolution.py:71-99 backward: grad_y [B][2F][T] -> g_kernel [F][2] (through impulse_responses.py:5-16 and the clamps
 * of convolution.py:15-22; nullable), g_x [B][T] (nullable). */
int leaf_gabor_conv_backward_f32(const float* x, int B, int T, const float* kernel, int F, int K, const float* grad_y,
                                 float* g_kernel, float* g_x, void* workspace, size_t workspace_bytes, void* stream);

/* frontend.py:15-19 backward: grad_y [B][2F][T] = 2 y grad_e (grad_e [B][F][T] broadcast over the re/im pair). */
int leaf_squared_modulus_backward_f32(const float* y, const float* grad_e, int B, int F, int T, float* grad_y,
                                      void* stream);

/* pooling.py:31-42 backward: grad_pooled [B][F][T'] -> g_e [B][F][T] (nullable), g_pool_w [F] (through
 * impulse_responses.py:74-80 incl. its clamp; nullable), g_pool_b [F] (nullable). */
int leaf_gaussian_lowpass_backward_f32(const float* e, const float* grad_pooled, int B, int F, int T,
                                       const float* pool_w, int K, int hop, float* g_e, float* g_pool_w,
                                       float* g_pool_b, void* workspace, size_t workspace_bytes, void* stream);

/* postprocessing.py:13-28 backward: grad_ema [B][F][T'] -> g_p [B][F][T'], g_ema_w [F] (per-channel coefficient; the
 * host sums it for a shared one). */
int leaf_ema_backward_f32(const float* p, const float* grad_ema, int B, int F, int TP, const float* ema_w, float* g_p,
                          float* g_ema_w, void* workspace, size_t workspace_bytes, void* stream);

/* postprocessing.py:62-69 backward (no 1e-5 floor in front: that one belongs to frontend.py:84). */
int leaf_pcen_backward_f32(const float* p, const float* grad_out, int B, int F, int TP, const float* alpha,
                           const float* delta, const float* root, const float* ema_w, float floor_, float* g_p,
                           float* g_alpha, float* g_delta, float* g_root, float* g_ema_w, void* workspace,
                           size_t workspace_bytes, void* stream);

/*
 * Inference with frozen parameters (serving): everything derived from (kernel, pool_w) alone -- the filter spectra and
 * the pooling rows of the overlap-save path -- is prepared once into a caller-owned `tables` buffer and reused by every
 * forward, which then skips the table kernel (7 us: 2 % of a 256-clip batch, 20 % of a 4-clip one).  The caller is
 * responsible for preparing again after the parameters change.  Same arithmetic, bit-identical outputs.
 * leaf_fft_tables_bytes returns 0 when the overlap-save path does not cover the geometry (use leaf_forward_f32).
 */
size_t leaf_fft_tables_bytes(int F, int K, int hop);
int leaf_fft_prepare_tables_f32(const float* kernel /*[F][2]*/, const float* pool_w /*[F]*/, int F, int K, int hop,
                                void* tables, size_t tables_bytes, void* stream);
/* x is float32, or bfloat16 with LEAF_FLAG_IO_BF16 (then out is bfloat16 too); workspace >= leaf_workspace_bytes(...,
 * LEAF_ALGO_FFT). */
int leaf_forward_prepared_f32(const float* x, int B, int T, const void* tables, size_t tables_bytes,
                              const float* pool_b, const float* alpha, const float* delta, const float* root,
                              const float* ema_w, int F, int K, int hop, int flags, float* out,
                              void* workspace, size_t workspace_bytes, void* stream);

/* utilities/data/raw_transforms.py:334-345 (PeakNormalization, apply_to="only_too_loud_sounds"; the last transform of
 * every reference data pipeline, there on the CPU through the third-party torch_audiomentations): clips whose peak |x|
 * exceeds 1 are divided by their peak, others are copied unchanged.  x, out [B][T]; out may alias x. */
int leaf_peak_normalize_f32(const float* x, int B, int T, float* out, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* LEAF_HIP_H_ */
