// leaf_backward.hpp -- backward kernels: staged transposes, PCEN/EMA reverse sweep, tap-gradient MFMA GEMM, chain to (mu, sigma)
// Part of the single translation unit leaf_kernels.hip (gfx950 only); see that file's header comment.
#pragma once
#include "leaf_common.hpp"
#include "leaf_fused.hpp"

namespace {

// ---------------------------------------------------------------------------------------------
// backward (staged, correctness-first): gradients of a scalar loss w.r.t. the seven parameters (and
// optionally x) given dL/d out.  Every forward intermediate is recomputed on the device with the staged
// kernels above; nothing is kept from the forward call.  Mirrors what autograd derives for the reference
// graph (frontend.py:78-89), including its clamp sub-gradients:
//   torch.clamp  -> gradient passes where lo <= x <= hi          (convolution.py:19-20, impulse_responses.py:75,
//                                                                  postprocessing.py:14)
//   torch.min/max against a scalar tensor -> the selected side; an exact tie splits 1/2 (postprocessing.py:63-64)
//   torch.maximum(p, 1e-5)               -> passes where p > 1e-5 (frontend.py:84)
// ---------------------------------------------------------------------------------------------

// One lane per (b,f) row.  raw = pooled before the floor.  Forward EMA is recomputed into `ema`, then the
// reverse-time sweep produces g_pre (grad w.r.t. raw) and the row's contributions to d alpha, d delta, d root,
// d ema_w in rowsum[row][4].  mode bit0: PCEN on; bit4 (16): no pooled floor (the stand-alone PCENLayer backward).
#ifndef LEAF_INST_TU               // non-template kernel: compiled once, in leaf_kernels.hip
__global__ void pcen_bwd_rows_kernel(const float* __restrict__ raw, const float* __restrict__ gout, int BF, int F, int TP,
                                     const float* __restrict__ alpha, const float* __restrict__ delta,
                                     const float* __restrict__ root, const float* __restrict__ ema_w, float floor_,
                                     int mode, float* __restrict__ ema, float* __restrict__ gpre,
                                     float* __restrict__ rowsum, const int* __restrict__ col_of, int FP,
                                     float* __restrict__ gcols) {
    const int row = blockIdx.x * blockDim.x + threadIdx.x;
    if (row >= BF) return;
    const float* r = raw + (size_t)row * TP;
    const float* go = gout + (size_t)row * TP;
    float* gp = gpre + (size_t)row * TP;
    const int f = row % F;
    // fused backward: also a [B][TP][FP] copy with filters in tap-column order
    float* gc = gcols ? gcols + (size_t)(row / F) * TP * FP + col_of[f] : nullptr;
    const bool nofloor = (mode & 16) != 0;          // stand-alone PCENLayer: no 1e-5 floor in front (frontend.py:84 is Leaf's)
    auto fl = [&](float v) { return nofloor ? v : fmaxf(v, kPooledFloor); };
    if (!(mode & 1)) {
        for (int m = 0; m < TP; ++m) {
            const float v = (nofloor || r[m] > kPooledFloor) ? go[m] : 0.0f;
            gp[m] = v;
            if (gc) gc[(size_t)m * FP] = v;
        }
        return;
    }
    float* M = ema + (size_t)row * TP;
    const float w = fminf(fmaxf(ema_w[f], 0.0f), 1.0f), omw = 1.0f - w;
    const float a = fminf(alpha[f], 1.0f);
    const float reff = fmaxf(root[f], 1.0f), rho = 1.0f / reff;
    const float d = delta[f];
    const float d_rho = powf(d, rho), ln_d = logf(d);
    const bool fast = d > 1e-30f;                                         // the hardware log2 / exp2 forms (see below)
    float state = fl(r[0]);
    for (int m = 0; m < TP; ++m) {
        const float p = fl(r[m]);
        state = w * p + omw * state;
        M[m] = state;
    }
    float s_a = 0.f, s_d = 0.f, s_rho = 0.f, s_w = 0.f, gM_next = 0.f;
    const float p0 = fl(r[0]);
    for (int m = TP - 1; m >= 0; --m) {
        const float p = fl(r[m]);
        const float Mf = floor_ + M[m];
        const float u = powf(Mf, a);
        const float v = p / u + d;
        const float vr = powf(v, rho);
        const float g = go[m];
        const float dv = rho * vr / v * g;
        s_d += dv - rho * d_rho / d * g;
        s_rho += (vr * logf(v) - d_rho * ln_d) * g;
        float dp = dv / u;
        const float du = -dv * p / (u * u);
        s_a += du * u * logf(Mf);
        const float gM = du * a * u / Mf + omw * gM_next;
        dp += w * gM;
        const float Mprev = m > 0 ? M[m - 1] : p0;
        s_w += gM * (p - Mprev);
        if (m == 0) dp += omw * gM;                 // the recurrence starts from p_0 (postprocessing.py:15)
        gM_next = gM;
        const float gv = (nofloor || r[m] > kPooledFloor) ? dp : 0.0f;
        gp[m] = gv;
        if (gc) gc[(size_t)m * FP] = gv;
    }
    const float al = alpha[f], ro = root[f], ew = ema_w[f];
    float* rs = rowsum + (size_t)row * 4;
    rs[0] = al < 1.0f ? s_a : (al == 1.0f ? 0.5f * s_a : 0.0f);
    rs[1] = s_d;
    const float g_reff = -s_rho * rho * rho;
    rs[2] = ro > 1.0f ? g_reff : (ro == 1.0f ? 0.5f * g_reff : 0.0f);
    rs[3] = (ew >= 0.0f && ew <= 1.0f) ? s_w : 0.0f;
}
#endif

// Same arithmetic as pcen_bwd_rows_kernel, one WAVE per (b,f) row instead of one lane: lane l owns frames 2l, 2l+1 of
// each 128-frame chunk.  The EMA (forward in time) and the gradient recurrence gM_m = c_m + (1-w) gM_{m+1} (backward in
// time) are first-order linear recurrences = compositions of affine maps: composed in-lane for the pair, scanned across
// the wavefront with 6 shuffle steps (up for the EMA, down for gM), with a carried state between chunks.  The serial
// kernel spends 137 us on B F = 10240 rows of 100 frames (160 waves, latency-bound); this one keeps the whole chip busy.
#ifndef LEAF_BWD_SCAN_REG
#define LEAF_BWD_SCAN_REG 1            // pcen_bwd_scan_kernel: rows of up to 512 frames keep the smoother's values in registers between its two passes; 0: through `ema` (A/B)
#endif
constexpr int kScanRegChunks = 4;
#ifndef LEAF_INST_TU               // non-template kernel: compiled once, in leaf_kernels.hip
__global__ __launch_bounds__(256) void pcen_bwd_scan_kernel(const float* __restrict__ raw, const float* __restrict__ gout, int BF,
                                                            int F, int TP, const float* __restrict__ alpha,
                                                            const float* __restrict__ delta, const float* __restrict__ root,
                                                            const float* __restrict__ ema_w, float floor_, int mode,
                                                            float* __restrict__ ema, float* __restrict__ gpre,
                                                            float* __restrict__ rowsum, const int* __restrict__ col_of, int FP,
                                                            float* __restrict__ gcols, float* __restrict__ grow = nullptr) {
    // grow (optional): grow[row] = sum_m g_pre[row][m], the row's share of d pool_b -- param_reduce_kernel then adds B numbers
    // per filter instead of re-reading the B x T' gradients (round 4: 22 -> 8 us at 256 x 1 s)
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (row >= BF) return;
    const float* r = raw + (size_t)row * TP;
    const float* go = gout + (size_t)row * TP;
    float* gp = gpre + (size_t)row * TP;
    const int f = row % F;
    float* gc = gcols ? gcols + (size_t)(row / F) * TP * FP + col_of[f] : nullptr;
    if (!(mode & 1)) {
        float sg = 0.0f;
        for (int m = lane; m < TP; m += 64) {
            const float v = r[m] > kPooledFloor ? go[m] : 0.0f;
            gp[m] = v;
            sg += v;
            if (gc) gc[(size_t)m * FP] = v;
        }
        if (grow) {
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) sg += __shfl_xor(sg, off);
            if (lane == 0) grow[row] = sg;
        }
        return;
    }
    float* M = ema + (size_t)row * TP;
    const float w = fminf(fmaxf(ema_w[f], 0.0f), 1.0f), omw = 1.0f - w;
    const float a = fminf(alpha[f], 1.0f);
    const float reff = fmaxf(root[f], 1.0f), rho = 1.0f / reff;
    const float d = delta[f];
    const float d_rho = powf(d, rho), ln_d = logf(d);
    const bool fast = d > 1e-30f;                                         // the hardware log2 / exp2 forms (see below)
    const float p0 = fmaxf(r[0], kPooledFloor);
    const int nchunk = (TP + 127) / 128;
    // rows of up to kScanRegChunks x 128 frames keep the smoother's values in registers between the two passes (the pass back in time
    // needs M_m and M_{m-1} of the lane's own two frames: a neighbour's register, not a round trip through `ema` in memory); longer
    // rows store them and read them back
    constexpr int RC = kScanRegChunks;
    const bool inreg = LEAF_BWD_SCAN_REG && nchunk <= RC;
    float M0r[RC], M1r[RC];
    // ---- forward in time: M_m = w p_m + (1-w) M_{m-1}, M_{-1} = p_0 (postprocessing.py:15)
    float carry = p0;
    auto fwd_chunk = [&](int c, float& M0, float& M1, bool& ok0, bool& ok1) {
        const int j0 = 128 * c + 2 * lane, j1 = j0 + 1;
        ok0 = j0 < TP;
        ok1 = j1 < TP;
        const float q0 = ok0 ? fmaxf(r[j0], kPooledFloor) : 0.0f, q1 = ok1 ? fmaxf(r[j1], kPooledFloor) : 0.0f;
        const float A0 = ok0 ? omw : 1.0f, B0 = ok0 ? w * q0 : 0.0f, A1 = ok1 ? omw : 1.0f, B1 = ok1 ? w * q1 : 0.0f;
        float A = A1 * A0, Bv = fmaf(A1, B0, B1);
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const float Ap = __shfl_up(A, off), Bp = __shfl_up(Bv, off);
            if (lane >= off) {
                Bv = fmaf(A, Bp, Bv);
                A *= Ap;
            }
        }
        const float Mend = fmaf(A, carry, Bv);
        const float Mprev = __shfl_up(Mend, 1);
        M0 = fmaf(A0, lane ? Mprev : carry, B0);
        M1 = fmaf(A1, M0, B1);
        carry = __shfl(Mend, 63);
    };
    if (inreg) {
#pragma unroll
        for (int c = 0; c < RC; ++c) {
            M0r[c] = M1r[c] = 0.0f;
            if (c < nchunk) {
                bool ok0, ok1;
                fwd_chunk(c, M0r[c], M1r[c], ok0, ok1);
            }
        }
    } else {
        for (int c = 0; c < nchunk; ++c) {
            float M0, M1;
            bool ok0, ok1;
            fwd_chunk(c, M0, M1, ok0, ok1);
            if (ok0) M[128 * c + 2 * lane] = M0;
            if (ok1) M[128 * c + 2 * lane + 1] = M1;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");           // M is re-read below by other lanes of this wave
        __builtin_amdgcn_s_waitcnt(0);
    }
    // ---- backward in time
    float s_a = 0.f, s_d = 0.f, s_rho = 0.f, s_w = 0.f, s_g = 0.f;
    float gnext = 0.0f;                                                   // gM of the first frame of the following chunk
    // Mv0 / Mv1: M at the lane's frames j0, j0 + 1; Mb: M at j0 - 1 (p_0 in front of the row)
    auto bwd_chunk = [&](int c, float Mv0, float Mv1, float Mb) {
        const int j0 = 128 * c + 2 * lane;
        float cm[2], dpd[2], pv[2], Mp[2];
        bool ok[2], above[2];
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const int m = j0 + k;
            ok[k] = m < TP;
            cm[k] = dpd[k] = pv[k] = Mp[k] = 0.0f;
            above[k] = false;
            if (ok[k]) {
                const float rv = r[m];
                const float p = fmaxf(rv, kPooledFloor);
                above[k] = rv > kPooledFloor;
                // powers and logarithms on the hardware log2 / exp2 (1 ulp each; the forward's leaf_pow_pos): one log2 serves both
                // the power and the logarithm of the same argument -- 2 v_log + 2 v_exp per frame instead of two powf and two
                // logf calls (~60-100 instructions each); non-positive v gives NaN / -inf exactly where powf / logf do
                // A filter whose learned delta is <= 0 (or denormal) keeps the library powf / logf: that is the function the forward
                // evaluates there (fin_point's literal form), v = p / u + delta may be tiny or non-positive, and the raw
                // v_log_f32 / v_exp_f32 do not handle denormals like powf.  Wave-uniform: delta is per row.
                const float Mf = floor_ + (k ? Mv1 : Mv0);
                float l2M, u, v, l2v, vr;
                if (fast) {
                    l2M = __builtin_amdgcn_logf(Mf);
                    u = __builtin_amdgcn_exp2f(a * l2M);
                    v = p / u + d;
                    l2v = __builtin_amdgcn_logf(v);
                    vr = __builtin_amdgcn_exp2f(rho * l2v);
                } else {
                    l2M = logf(Mf) * 1.4426950408889634f;
                    u = powf(Mf, a);
                    v = p / u + d;
                    l2v = logf(v) * 1.4426950408889634f;
                    vr = powf(v, rho);
                }
                const float g = go[m];
                const float dv = rho * vr / v * g;
                s_d += dv - rho * d_rho / d * g;
                s_rho += (vr * (l2v * 0.6931471805599453f) - d_rho * ln_d) * g;
                dpd[k] = dv / u;
                const float du = -dv * p / (u * u);
                s_a += du * u * (l2M * 0.6931471805599453f);
                cm[k] = du * a * u / Mf;
                pv[k] = p;
                Mp[k] = k ? Mv0 : Mb;
            }
        }
        // gM_m = cm_m + beta_m gM_{m+1}: pair map, then the inclusive scan over lanes from the high end
        const float b0 = ok[0] ? omw : 1.0f, b1 = ok[1] ? omw : 1.0f;
        float A = b0 * b1, Bv = fmaf(b0, cm[1], cm[0]);
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const float Ap = __shfl_down(A, off), Bp = __shfl_down(Bv, off);
            if (lane + off < 64) {
                Bv = fmaf(A, Bp, Bv);
                A *= Ap;
            }
        }
        const float g0 = fmaf(A, gnext, Bv);                              // gM of this lane's first frame
        const float gn = __shfl_down(g0, 1);
        const float g1 = fmaf(b1, lane < 63 ? gn : gnext, cm[1]);         // gM of this lane's second frame
        gnext = __shfl(g0, 0);
        const float gM[2] = {g0, g1};
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const int m = j0 + k;
            if (ok[k]) {
                float dp = dpd[k] + w * gM[k];
                s_w += gM[k] * (pv[k] - Mp[k]);
                if (m == 0) dp += omw * gM[k];                            // the recurrence starts from p_0
                const float gv = above[k] ? dp : 0.0f;
                gp[m] = gv;
                s_g += gv;
                if (gc) gc[(size_t)m * FP] = gv;
            }
        }
    };
    if (inreg) {
#pragma unroll
        for (int c = RC - 1; c >= 0; --c) {
            // M at j0 - 1: the previous lane's second frame; lane 0: the previous chunk's last frame (p_0 in front of the row)
            const float up = __shfl_up(M1r[c], 1);
            const float prev_chunk = c > 0 ? __shfl(M1r[c > 0 ? c - 1 : 0], 63) : p0;
            if (c < nchunk) bwd_chunk(c, M0r[c], M1r[c], lane ? up : prev_chunk);
        }
    } else {
        for (int c = nchunk - 1; c >= 0; --c) {
            const int j0 = 128 * c + 2 * lane;
            const float Mv0 = j0 < TP ? M[j0] : 0.0f, Mv1 = j0 + 1 < TP ? M[j0 + 1] : 0.0f;
            const float Mb = (j0 < TP && j0 > 0) ? M[j0 - 1] : p0;
            bwd_chunk(c, Mv0, Mv1, Mb);
        }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        s_a += __shfl_xor(s_a, off);
        s_d += __shfl_xor(s_d, off);
        s_rho += __shfl_xor(s_rho, off);
        s_w += __shfl_xor(s_w, off);
        s_g += __shfl_xor(s_g, off);
    }
    if (lane == 0) {
        if (grow) grow[row] = s_g;
        const float al = alpha[f], ro = root[f], ew = ema_w[f];
        float* rs = rowsum + (size_t)row * 4;
        rs[0] = al < 1.0f ? s_a : (al == 1.0f ? 0.5f * s_a : 0.0f);
        rs[1] = s_d;
        const float g_reff = -s_rho * rho * rho;
        rs[2] = ro > 1.0f ? g_reff : (ro == 1.0f ? 0.5f * g_reff : 0.0f);
        rs[3] = (ew >= 0.0f && ew <= 1.0f) ? s_w : 0.0f;
    }
}
#endif

// d e[b,f,n] = sum_m g[f][n + padL - m hop] * gpre[b,f,m]  (transpose of pooling.py:41), then
// dy[b,2f,n] = 2 y_re de, dy[b,2f+1,n] = 2 y_im de written over y  (frontend.py:15-19).
#ifndef LEAF_INST_TU               // non-template kernel: compiled once, in leaf_kernels.hip
__global__ void pool_bwd_dy_kernel(float* __restrict__ y, const float* __restrict__ g, const float* __restrict__ gpre,
                                   int F, int T, int TP, int K, int hop, int padL) {
    const int n = blockIdx.x * blockDim.x + threadIdx.x;
    const int f = blockIdx.y, b = blockIdx.z;
    if (n >= T) return;
    const float* w = g + (size_t)f * K;
    const float* gp = gpre + ((size_t)b * F + f) * TP;
    const int np = n + padL;
    const int m_hi = min(TP - 1, np / hop);
    const int m_lo = max(0, (np - K + hop) / hop);          // smallest m with np - m*hop <= K-1
    float de = 0.0f;
    for (int m = m_lo; m <= m_hi; ++m) {
        const int j = np - m * hop;
        if (j >= 0 && j < K) de = fmaf(w[j], gp[m], de);
    }
    const size_t ire = ((size_t)b * 2 * F + 2 * f) * T + n;
    y[ire] *= 2.0f * de;
    y[ire + T] *= 2.0f * de;
}
#endif

// dg[f][j] = sum_{b,m} gpre[b,f,m] * ez[b,f,m hop + j - padL]
#ifndef LEAF_INST_TU               // non-template kernel: compiled once, in leaf_kernels.hip
__global__ void pool_bwd_dg_kernel(const float* __restrict__ e, const float* __restrict__ gpre, int B, int F, int T, int TP,
                                   int K, int hop, int padL, float* __restrict__ dg) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    const int f = blockIdx.y;
    if (j >= K) return;
    float acc = 0.0f;
    for (int b = 0; b < B; ++b) {
        const float* eb = e + ((size_t)b * F + f) * T;
        const float* gp = gpre + ((size_t)b * F + f) * TP;
        for (int m = 0; m < TP; ++m) {
            const int n = m * hop + j - padL;
            if (n >= 0 && n < T) acc = fmaf(gp[m], eb[n], acc);
        }
    }
    dg[(size_t)f * K + j] = acc;
}
#endif

// One block per filter: d pool_b, d pool_w and the PCEN parameter sums over the batch.
constexpr int kParamRedThreads = 1024;
#ifndef LEAF_PARAM_REDUCE_BATCHED
#define LEAF_PARAM_REDUCE_BATCHED 1    // param_reduce_kernel: the eight sums of the overlap-save backwards loaded together, one barrier; 0: one after the other (A/B, same bits)
#endif
#ifndef LEAF_INST_TU               // non-template kernel: compiled once, in leaf_kernels.hip
__global__ __launch_bounds__(kParamRedThreads) void param_reduce_kernel(const float* __restrict__ gpre, const float* __restrict__ dg,
                                    const float* __restrict__ g, const float* __restrict__ rowsum,
                                    const float* __restrict__ pool_w, int B, int F, int TP, int K, int mode,
                                    const float* __restrict__ dwpart, int dw_rows, int FP,
                                    const int* __restrict__ col_of, float* __restrict__ g_pool_w, float* __restrict__ g_pool_b, float* __restrict__ g_alpha,
                                    float* __restrict__ g_delta, float* __restrict__ g_root, float* __restrict__ g_ema,
                                    const float* __restrict__ dkpart = nullptr, int dk_blocks = 0, const float* __restrict__ kernel = nullptr,
                                    GaborBounds bd = GaborBounds{}, float* __restrict__ g_kernel = nullptr,
                                    const float* __restrict__ grow = nullptr) {
    __shared__ float red[2][kParamRedThreads / 64];
    const int f = blockIdx.x, tid = threadIdx.x;
    // sums of the block in a FIXED order (bit-reproducible): DPP / shuffle tree inside each wave, the sixteen wave sums added in wave
    // order by every thread -- two barriers per sum instead of the eleven of a shared-memory tree (round 4: this kernel is one of the
    // small launches around the main backward kernel; its seven sums were ~6 us of barriers)
    int parity = 0;
    auto block_sum = [&](float v) {
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
        float* r = red[parity];
        parity ^= 1;                                  // two buffers: the next sum's writes cannot overtake this sum's reads
        if ((tid & 63) == 0) r[tid >> 6] = v;
        __syncthreads();
        float t = 0.0f;
#pragma unroll
        for (int w = 0; w < kParamRedThreads / 64; ++w) t += r[w];
        return t;
    };
    if (LEAF_PARAM_REDUCE_BATCHED && grow && dwpart) {
        // The overlap-save backwards (rows' sums from the scan kernel, per-block partials from the main kernel): every load of all
        // eight sums first, four strides of each at a time, then ONE barrier for the eight block sums -- one memory latency and one
        // barrier instead of eight of each (this kernel is a launch of a few microseconds between the main kernel and the caller).
        // Every sum adds the same numbers in the same order as the sequential form below (a lane's strided values ascending, the
        // wave's xor tree, the sixteen waves in order): the same bits.
        __shared__ float red8[kParamRedThreads / 64][8];
        const int col = col_of[f];
        const bool pc = (mode & 1) != 0;
        const int ndk = dkpart ? dk_blocks : 0;
        const int nmax = max(max(B, dw_rows), ndk);
        float v[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        for (int i0 = tid; i0 < nmax; i0 += 4 * kParamRedThreads) {
            float gr[4], dw[4];
            float4 rs[4];
            float2 dk[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int i = i0 + u * kParamRedThreads;
                gr[u] = i < B ? grow[(size_t)i * F + f] : 0.0f;
                rs[u] = (pc && i < B) ? *reinterpret_cast<const float4*>(rowsum + ((size_t)i * F + f) * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
                dw[u] = i < dw_rows ? dwpart[(size_t)i * FP + col] : 0.0f;
                dk[u] = i < ndk ? *reinterpret_cast<const float2*>(dkpart + ((size_t)i * F + f) * 2) : make_float2(0.f, 0.f);
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {                     // (adding +0 for a stride past an array's end leaves the sum as it is)
                v[0] += gr[u];
                v[1] += dw[u];
                v[2] += rs[u].x;
                v[3] += rs[u].y;
                v[4] += rs[u].z;
                v[5] += rs[u].w;
                v[6] += dk[u].x;
                v[7] += dk[u].y;
            }
        }
#pragma unroll
        for (int q = 0; q < 8; ++q) {
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) v[q] += __shfl_xor(v[q], off);
        }
        if ((tid & 63) == 0) {
#pragma unroll
            for (int q = 0; q < 8; ++q) red8[tid >> 6][q] = v[q];
        }
        __syncthreads();
        if (tid == 0) {
            float t[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                t[q] = 0.0f;
#pragma unroll
                for (int w = 0; w < kParamRedThreads / 64; ++w) t[q] += red8[w][q];
            }
            const float wr0 = pool_w[f];
            if (g_pool_b) g_pool_b[f] = t[0];
            g_pool_w[f] = (wr0 >= 2.0f / (float)K && wr0 <= 0.5f) ? t[1] : 0.0f;
            if (pc) {
                g_alpha[f] = t[2];
                g_delta[f] = t[3];
                g_root[f] = t[4];
                g_ema[f] = t[5];
            }
            if (dkpart) {
                const float mu_raw = kernel[2 * f], sg_raw = kernel[2 * f + 1];
                g_kernel[2 * f] = (mu_raw >= 0.0f && mu_raw <= 3.14159274101257324f) ? t[6] : 0.0f;
                g_kernel[2 * f + 1] = (sg_raw >= bd.sigma_lo && sg_raw <= bd.sigma_hi) ? t[7] : 0.0f;
            }
        }
        return;
    }
    float acc = 0.0f;
    if (grow) {                                       // the rows' sums (pcen_bwd_scan_kernel)
        for (int b = tid; b < B; b += kParamRedThreads) acc += grow[(size_t)b * F + f];
    } else {
        for (int i = tid; i < B * TP; i += kParamRedThreads) {
            const int b = i / TP, m = i - b * TP;
            acc += gpre[((size_t)b * F + f) * TP + m];
        }
    }
    const float sb = block_sum(acc);
    // d g/d s = g * (j - c)^2 / (c^2 s^3), c = (K-1)/2   (impulse_responses.py:75-80)
    const float wr = pool_w[f];
    const float sig = pool_sigma(wr, K);
    const float c = 0.5f * (float)(K - 1);
    acc = 0.0f;
    if (dwpart) {                                     // fused backward: per-wave partial sums, tap-column order
        const int col = col_of[f];
        for (int i = tid; i < dw_rows; i += kParamRedThreads) acc += dwpart[(size_t)i * FP + col];
    } else {
        for (int j = tid; j < K; j += kParamRedThreads) {
            const float t = (float)j - c;
            acc += dg[(size_t)f * K + j] * g[(size_t)f * K + j] * (t * t) / (c * c * sig * sig * sig);
        }
    }
    const float sw = block_sum(acc);
    float sums[4] = {0.f, 0.f, 0.f, 0.f};
    if (mode & 1) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            acc = 0.0f;
            for (int b = tid; b < B; b += kParamRedThreads) acc += rowsum[((size_t)b * F + f) * 4 + q];
            sums[q] = block_sum(acc);
        }
    }
    // (overlap-save backward) the per-block (d mu, d sigma) partials of this filter and the clamp sub-gradients of
    // convolution.py:15-22 -- what fft_dkernel_reduce_kernel did in a launch of its own
    if (dkpart) {
        float a = 0.0f, c2 = 0.0f;
        for (int i = tid; i < dk_blocks; i += kParamRedThreads) {
            a += dkpart[((size_t)i * F + f) * 2];
            c2 += dkpart[((size_t)i * F + f) * 2 + 1];
        }
        const float smu = block_sum(a), ssg = block_sum(c2);
        if (tid == 0) {
            const float mu_raw = kernel[2 * f], sg_raw = kernel[2 * f + 1];
            g_kernel[2 * f] = (mu_raw >= 0.0f && mu_raw <= 3.14159274101257324f) ? smu : 0.0f;
            g_kernel[2 * f + 1] = (sg_raw >= bd.sigma_lo && sg_raw <= bd.sigma_hi) ? ssg : 0.0f;
        }
    }
    if (tid == 0) {
        if (g_pool_b) g_pool_b[f] = sb;
        g_pool_w[f] = (wr >= 2.0f / (float)K && wr <= 0.5f) ? sw : 0.0f;
        if (mode & 1) {
            g_alpha[f] = sums[0];
            g_delta[f] = sums[1];
            g_root[f] = sums[2];
            g_ema[f] = sums[3];
        }
    }
}
#endif

// dtaps partial per clip: part[b][c][j] = sum_n dy[b,c,n] * xz[b, n + j - padL]   (transpose of convolution.py:97 w.r.t. weights)
#ifndef LEAF_INST_TU               // non-template kernel: compiled once, in leaf_kernels.hip
__global__ void dtaps_partial_kernel(const float* __restrict__ dy, const float* __restrict__ x, int T, int C, int K, int padL,
                                     float* __restrict__ part) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    const int c = blockIdx.y, b = blockIdx.z;
    if (j >= K) return;
    const float* d = dy + ((size_t)b * C + c) * T;
    const float* xb = x + (size_t)b * T;
    const int off = j - padL;
    const int n0 = max(0, -off), n1 = min(T, T - off);
    float acc = 0.0f;
    for (int n = n0; n < n1; ++n) acc = fmaf(d[n], xb[n + off], acc);
    part[((size_t)b * C + c) * K + j] = acc;
}
#endif

// One block per filter: sum the per-clip tap gradients over the batch and chain them through the Gabor formula
// (impulse_responses.py:5-16) to (mu, sigma):  d hr/d mu = -t hi, d hi/d mu = t hr, d h/d sigma = h (t^2/s^3 - 1/s).
#ifndef LEAF_INST_TU               // non-template kernel: compiled once, in leaf_kernels.hip
__global__ void dkernel_kernel(const float* __restrict__ part, const float* __restrict__ taps,
                               const float* __restrict__ kernel, int B, int F, int K, GaborBounds bd,
                               float* __restrict__ g_kernel) {
    __shared__ float red[256];
    const int f = blockIdx.x, tid = threadIdx.x;
    const float mu_raw = kernel[2 * f], sg_raw = kernel[2 * f + 1];
    const float sg = fminf(fmaxf(sg_raw, bd.sigma_lo), bd.sigma_hi);
    float a_mu = 0.0f, a_sg = 0.0f;
    for (int j = tid; j < K; j += 256) {
        float dre = 0.0f, dim = 0.0f;
        for (int b = 0; b < B; ++b) {
            dre += part[((size_t)b * 2 * F + 2 * f) * K + j];
            dim += part[((size_t)b * 2 * F + 2 * f + 1) * K + j];
        }
        const float t = (float)(j - K / 2);
        const float hr = taps[(size_t)(2 * f) * K + j], hi = taps[(size_t)(2 * f + 1) * K + j];
        a_mu += t * (dim * hr - dre * hi);
        a_sg += (dre * hr + dim * hi) * (t * t / (sg * sg * sg) - 1.0f / sg);
    }
    float out2[2];
    float vals[2] = {a_mu, a_sg};
    for (int q = 0; q < 2; ++q) {
        red[tid] = vals[q];
        __syncthreads();
        for (int s = 128; s > 0; s >>= 1) {
            if (tid < s) red[tid] += red[tid + s];
            __syncthreads();
        }
        out2[q] = red[0];
        __syncthreads();
    }
    if (tid == 0) {
        g_kernel[2 * f] = (mu_raw >= 0.0f && mu_raw <= 3.14159274101257324f) ? out2[0] : 0.0f;
        g_kernel[2 * f + 1] = (sg_raw >= bd.sigma_lo && sg_raw <= bd.sigma_hi) ? out2[1] : 0.0f;
    }
}
#endif

// dx[b,i] = sum_c sum_j taps[c][j] * dy[b,c,i - j + padL]
#ifndef LEAF_INST_TU               // non-template kernel: compiled once, in leaf_kernels.hip
__global__ void dx_kernel(const float* __restrict__ dy, const float* __restrict__ taps, int T, int C, int K, int padL,
                          float* __restrict__ dx) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const int b = blockIdx.y;
    if (i >= T) return;
    float acc = 0.0f;
    for (int c = 0; c < C; ++c) {
        const float* d = dy + ((size_t)b * C + c) * T;
        const float* w = taps + (size_t)c * K;
        const int j0 = max(0, i + padL - (T - 1)), j1 = min(K, i + padL + 1);
        for (int j = j0; j < j1; ++j) acc = fmaf(w[j], d[i + padL - j], acc);
    }
    dx[(size_t)b * T + i] = acc;
}
#endif

// ---------------------------------------------------------------------------------------------
// fused backward, phase C: tap gradients as an fp32-MFMA GEMM.
//   dH[kk][c] = sum_{b,n} S_kk[b,n] * dY[b,n][c]   (c < FP, Re columns)      S_kk[n] = x[n+kk] + x[n-kk]
//   dH[kk][c] = sum_{b,n} D_kk[b,n] * dY[b,n][c]   (c >= FP, Im columns)     D_kk[n] = x[n+kk] - x[n-kk]
// i.e. the transpose of the forward GEMMs w.r.t. the tap table W, with the same Hermitian operands built from an
// LDS waveform window.  Rows = 16 tap rows per wave (one k-tile each), columns = the group's 16-filter tiles,
// reduction = time.  A workgroup walks 64-sample chunks (waveform window + dY tile double-buffered in LDS, next
// chunk prefetched into registers under the MFMAs) and finally writes its partial dH; a small kernel sums the
// partials and chains them to (mu, sigma).
// ---------------------------------------------------------------------------------------------
struct DtapsParams {
    const float* x;        // [B][T]
    const float* dY;       // [B*T][2*FP]
    const int* tile_ks;    // k-steps per column tile (support-sorted)
    float* dHpart;         // [gridDim.x][16*NKT][2*FP]
    int B, T, FP, K, Hf, xshift;
    int NKT;               // 16-row k-tiles
    int NW;                // waves per workgroup
    int NS;                // samples per chunk (multiple of 16)
    int HPc;               // window halo = 16*NKT
    int XSC;               // window floats = NS + 2*HPc
    int LD;                // LDS row stride of the dY tile = 2*FP + 16
    int nch;               // chunks per clip
    int total_chunks;      // B * nch
    int tile_base;         // first column tile of this launch's group 0
};
constexpr unsigned leaf_layout_hash_bwd() {
    return leaf_mix(leaf_mix(leaf_layout_hash_fused(), sizeof(DtapsParams)), offsetof(DtapsParams, tile_base));
}

template <int RT, int NA, bool EVENK>
__device__ __forceinline__ void dtaps_ktile(f32x4 (&acc)[2 * RT], const float* xc, const float* sdy, int LD, int colre,
                                            int colim, int krow, int g, int Hf, int NS) {
    for (int nb = 0; nb < NS / 16; ++nb) {
#pragma unroll
        for (int s4 = 0; s4 < 4; ++s4) {
            const int rr = 16 * nb + 4 * s4 + g;
            float fw = xc[rr + krow];
            const float bw = xc[rr - krow];
            if (EVENK) fw = krow <= Hf ? fw : 0.0f;
            const float sv = fw + bw, dv = fw - bw;
            const float* row = sdy + rr * LD;
#pragma unroll
            for (int t = 0; t < NA; ++t) {
                acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(sv, row[colre + 16 * t], acc[t], 0, 0, 0);
                acc[RT + t] = __builtin_amdgcn_mfma_f32_16x16x4f32(dv, row[colim + 16 * t], acc[RT + t], 0, 0, 0);
            }
        }
    }
}

constexpr int kDtPF = 4;      // float4 registers per thread for the dY tile prefetch
template <int RT, int TPW, bool EVENK>
__global__ __launch_bounds__(1024) void dtaps_mfma_kernel(const DtapsParams p) {
    extern __shared__ __attribute__((aligned(16))) float dsm[];
    const int tid = threadIdx.x, nthreads = blockDim.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    const int li = lane & 15, g = lane >> 4;
    const int tile_floats = p.NS * p.LD;
    const int buf_floats = (p.XSC + 3) / 4 * 4 + tile_floats;
    const int tile0 = p.tile_base + blockIdx.y * RT;
    const int colre = 16 * tile0 + li, colim = p.FP + 16 * tile0 + li;
    const int row4 = 2 * p.FP / 4;                       // float4 per dY row
    const int n4 = p.NS * row4;                          // float4 per dY tile

    // k-tile -> (wave, slot) assignment.  Column tiles are support-sorted, so low k-tiles carry more MFMAs (all column
    // tiles reach them) than high ones: a round-robin split leaves one SIMD with ~30 % more work.  Thread 0 does a
    // longest-processing-time greedy that balances the four SIMDs (waves w, w+4, .. share SIMD w & 3).
    __shared__ int s_kt[16 * 3];
    if (tid == 0) {
        int load[16], cnt[16], simd_load[4] = {0, 0, 0, 0};
        for (int w = 0; w < 16; ++w) load[w] = cnt[w] = 0;
        for (int i = 0; i < 16 * 3; ++i) s_kt[i] = -1;
        for (int kt = 0; kt < p.NKT; ++kt) {
            int work = 0;
            for (int t = 0; t < RT; ++t) work += (4 * p.tile_ks[tile0 + t] > 16 * kt) ? 1 : 0;
            if (work == 0) continue;
            int best = -1, best_key = 1 << 30;
            for (int w = 0; w < p.NW; ++w) {
                if (cnt[w] >= TPW) continue;
                const int key = simd_load[w & 3] * 64 + load[w];
                if (key < best_key) { best_key = key; best = w; }
            }
            s_kt[best * TPW + cnt[best]] = kt;
            cnt[best]++; load[best] += work; simd_load[best & 3] += work;
        }
    }
    __syncthreads();
    int na[TPW], ktile[TPW];                              // owned k-tiles and their active column tiles
#pragma unroll
    for (int tp = 0; tp < TPW; ++tp) {
        const int kt = __builtin_amdgcn_readfirstlane(s_kt[wave * TPW + tp]);
        ktile[tp] = kt;
        int n = 0;
        for (int t = 0; t < RT; ++t) n += (kt >= 0 && 4 * p.tile_ks[tile0 + t] > 16 * kt) ? 1 : 0;
        na[tp] = __builtin_amdgcn_readfirstlane(n);
    }
    f32x4 acc[TPW][2 * RT];
#pragma unroll
    for (int tp = 0; tp < TPW; ++tp)
#pragma unroll
        for (int c = 0; c < 2 * RT; ++c) acc[tp][c] = f32x4{0.f, 0.f, 0.f, 0.f};

    f32x4 pre[kDtPF];
    float prex[2];
    auto load_chunk = [&](int chunk) {
        const int b = chunk / p.nch, n0 = (chunk - b * p.nch) * p.NS;
        const float* src = p.dY + ((size_t)b * p.T + n0) * (size_t)(2 * p.FP);
#pragma unroll
        for (int i = 0; i < kDtPF; ++i) {
            const int idx = tid + i * nthreads;
            pre[i] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (idx < n4 && n0 + idx / row4 < p.T) pre[i] = *reinterpret_cast<const f32x4*>(src + (size_t)idx * 4);
        }
        const float* xb = p.x + (size_t)b * p.T;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int idx = tid + i * nthreads;
            const int n = n0 - p.HPc + p.xshift + idx;
            prex[i] = (idx < p.XSC && n >= 0 && n < p.T) ? xb[n] : 0.0f;
        }
    };
    auto store_chunk = [&](float* buf) {
        float* xw = buf;
        float* sdy = buf + (p.XSC + 3) / 4 * 4;
#pragma unroll
        for (int i = 0; i < kDtPF; ++i) {
            const int idx = tid + i * nthreads;
            if (idx < n4) {
                const int row = idx / row4, c4 = idx - row * row4;
                *reinterpret_cast<f32x4*>(sdy + row * p.LD + 4 * c4) = pre[i];
            }
        }
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int idx = tid + i * nthreads;
            if (idx < p.XSC) xw[idx] = prex[i];
        }
    };

    int chunk = blockIdx.x;
    int cur = 0;
    if (chunk < p.total_chunks) {
        load_chunk(chunk);
        store_chunk(dsm);
    }
    __syncthreads();
    for (; chunk < p.total_chunks; chunk += gridDim.x) {
        const int next = chunk + gridDim.x;
        if (next < p.total_chunks) load_chunk(next);
        const float* buf = dsm + (size_t)cur * buf_floats;
        const float* xc = buf + p.HPc;
        const float* sdy = buf + (p.XSC + 3) / 4 * 4;
#pragma unroll
        for (int tp = 0; tp < TPW; ++tp) {
            const int krow = 16 * ktile[tp] + li;
            if (na[tp] == RT) dtaps_ktile<RT, RT, EVENK>(acc[tp], xc, sdy, p.LD, colre, colim, krow, g, p.Hf, p.NS);
            if constexpr (RT >= 2)
                if (na[tp] == RT - 1)
                    dtaps_ktile<RT, RT - 1, EVENK>(acc[tp], xc, sdy, p.LD, colre, colim, krow, g, p.Hf, p.NS);
            if constexpr (RT >= 3)
                if (na[tp] == RT - 2)
                    dtaps_ktile<RT, RT - 2, EVENK>(acc[tp], xc, sdy, p.LD, colre, colim, krow, g, p.Hf, p.NS);
        }
        if (next < p.total_chunks) store_chunk(dsm + (size_t)(cur ^ 1) * buf_floats);
        __syncthreads();
        cur ^= 1;
    }
    // partial dH of this workgroup: D layout row = 4g + r (tap row within the k-tile), col = li
    float* outp = p.dHpart + (size_t)blockIdx.x * (16 * p.NKT) * (2 * p.FP);
#pragma unroll
    for (int tp = 0; tp < TPW; ++tp) {
        const int kt = ktile[tp];
        if (kt < 0) continue;
#pragma unroll
        for (int t = 0; t < RT; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const size_t rowoff = (size_t)(16 * kt + 4 * g + r) * (2 * p.FP);
                outp[rowoff + colre + 16 * t] = acc[tp][t][r];
                outp[rowoff + colim + 16 * t] = acc[tp][RT + t][r];
            }
    }
}

// sum the per-workgroup partial dH slabs: out[i] = sum_w part[w][i]
#ifndef LEAF_INST_TU               // non-template kernel: compiled once, in leaf_kernels.hip
__global__ void dh_reduce_kernel(const float* __restrict__ part, int nparts, size_t n, float* __restrict__ out) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float acc = 0.0f;
    for (int w = 0; w < nparts; ++w) acc += part[(size_t)w * n + i];
    out[i] = acc;
}
#endif

// One block per filter: sum the workgroup partials of dH and chain through the Gabor formula using the tap table
// itself (W = h * scale, and the scale cancels): d mu = sum_kk kk (dH_im W_re - dH_re W_im),
// d sigma = sum_kk (dH_re W_re + dH_im W_im) (kk^2/s^3 - 1/s); clamp sub-gradients as torch.clamp.
#ifndef LEAF_INST_TU               // non-template kernel: compiled once, in leaf_kernels.hip
__global__ void dkernel_fused_kernel(const float* __restrict__ dHpart, int nparts, int Rp, const float* __restrict__ W,
                                     int R, int FP, const int* __restrict__ col_of, const float* __restrict__ kernel,
                                     int F, GaborBounds bd, float* __restrict__ g_kernel) {
    (void)F;
    __shared__ float red[256];
    const int f = blockIdx.x, tid = threadIdx.x;
    const int c = col_of[f];
    const float mu_raw = kernel[2 * f], sg_raw = kernel[2 * f + 1];
    const float sg = fminf(fmaxf(sg_raw, bd.sigma_lo), bd.sigma_hi);
    float a_mu = 0.0f, a_sg = 0.0f;
    for (int kk = tid; kk < R; kk += 256) {
        float dre = 0.0f, dim = 0.0f;
        for (int w = 0; w < nparts; ++w) {
            const float* row = dHpart + ((size_t)w * Rp + kk) * (2 * FP);
            dre += row[c];
            dim += row[FP + c];
        }
        const float wre = W[(size_t)kk * (2 * FP) + c], wim = W[(size_t)kk * (2 * FP) + FP + c];
        const float t = (float)kk;
        a_mu += t * (dim * wre - dre * wim);
        a_sg += (dre * wre + dim * wim) * (t * t / (sg * sg * sg) - 1.0f / sg);
    }
    float res[2];
    const float vals[2] = {a_mu, a_sg};
    for (int q = 0; q < 2; ++q) {
        red[tid] = vals[q];
        __syncthreads();
        for (int s2 = 128; s2 > 0; s2 >>= 1) {
            if (tid < s2) red[tid] += red[tid + s2];
            __syncthreads();
        }
        res[q] = red[0];
        __syncthreads();
    }
    if (tid == 0) {
        g_kernel[2 * f] = (mu_raw >= 0.0f && mu_raw <= 3.14159274101257324f) ? res[0] : 0.0f;
        g_kernel[2 * f + 1] = (sg_raw >= bd.sigma_lo && sg_raw <= bd.sigma_hi) ? res[1] : 0.0f;
    }
}
#endif

}  // namespace
