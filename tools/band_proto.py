#!/usr/bin/env python3
"""Numerical prototype (CPU, numpy fp64) of the band-limited filter tasks (round 5; DESIGN.md "what comes next" of round 4).

For a narrow-band Gabor filter the block's 2048-point inverse transform is replaced by an M-point inverse (M = 256, 512) of
the M bins around the filter's centre: z[m] = y_bl(n_c + m D), D = 2048 / M, up to a phase that |.|^2 removes.  The pooled
frame sums are then taken at the decimated rate with the window G~(tau) = D (g * phi)(tau), phi a low-pass with cutoff
1 / (2 D) (Kaiser-windowed sinc) that removes the window's spectrum around the multiples of M (the un-normalised Gaussian
window is TRUNCATED: its edge step has 1/k tails).  Block boundaries cut the DECIMATED sequence (an exact partition of the
sum); frames cut by the clip's ends -- or whose phi tails would reach past them -- use per-block circular tables
(W * phi, circular), which are exact for the block's periodic band-limited signal.

The class of a filter is decided from its own spectrum R (as the device does, from the table it has just built):
    out-of-window energy  sum_{k outside} R^2  <=  eps^2 sum R^2                 (what the short transform drops)
    |autocorrelation of R at lags M/2, 3M/4|  <=  eta sum R^2                    (|y|^2 content that would alias)
This script measures, per filter, the error of the pooled sums of the chosen class against the exact fp64 reference graph.
   usage: band_proto.py [--sr 16000] [--lh 12] [--beta 12] [--seed 0] [--fuzz N]"""
import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

N = 2048
CLASSES = (256, 512)


def default_kernel(F, sr, K):
    from leaf_pytorch_amd.initializers import GaborInit
    return GaborInit(default_window_len=K, sample_rate=sr, min_freq=60.0, max_freq=7800.0)((F, 2)).numpy().astype(np.float64)


def taps(kern, K):
    c = np.sqrt(2 * np.log(2)) / np.pi
    t = np.arange(-(K // 2), (K + 1) // 2)
    mu = np.clip(kern[:, 0:1], 0, np.pi)
    sg = np.clip(kern[:, 1:2], 4 * c, K * c)
    return np.exp(-t ** 2 / (2 * sg ** 2)) * np.exp(1j * mu * t) / (np.sqrt(2 * np.pi) * sg)


def exact(x, h, g, hop):
    """y[n] = sum_j h[j] x[n + j - padL]; e = |y|^2 on [0, T); p[m] = sum_j g[j] e[m hop + j - padL]"""
    F, K = h.shape
    T = x.shape[0]
    padl = K // 2 + K % 2 - 1
    L = 1 << int(np.ceil(np.log2(T + 2 * K)))
    X = np.fft.fft(np.concatenate([np.zeros(padl), x, np.zeros(L - T - padl)]))
    TP = (T - 1) // hop + 1
    p = np.zeros((F, TP))
    for f in range(F):
        Hn = np.fft.fft(h[f], L)
        Hn = np.concatenate([Hn[:1], Hn[:0:-1]])        # H[-k]
        y = np.fft.ifft(X * Hn)[:T]
        e = np.abs(y) ** 2
        ez = np.concatenate([np.zeros(padl), e, np.zeros(K)])
        idx = np.arange(TP)[:, None] * hop + np.arange(K)[None, :]
        p[f] = ez[idx] @ g[f]
    return p


def kaiser_lowpass(D, lh, beta):
    """odd-length windowed sinc with cutoff 1 / (2 D) cycles per sample, unit DC gain; lh = half length in DECIMATED samples"""
    n = np.arange(-lh * D, lh * D + 1)
    phi = np.sinc(n / D) / D * np.kaiser(len(n), beta)
    return phi / phi.sum(), lh * D


FWD_BINS = 1152


def window_start(k0, M):
    """first bin of the M-bin window inside the half spectrum 0..1024 (bins counted on the side the filter lives on)"""
    return int(min(max(k0 - M // 2, 0), N // 2 + 1 - M))


def decide(Rabs, k0, eps, eta, classes=None):
    classes = classes or CLASSES
    """Rabs: |spectrum| on bins 0..2047 with the filter's peak at k0 in 0..1024.  Returns (M, kb) or (2048, 0)."""
    tot = float((Rabs ** 2).sum())
    for M in classes:
        kb = window_start(k0, M)
        w = Rabs[kb:kb + M]
        out2 = float((Rabs[:kb] ** 2).sum() + (Rabs[kb + M:] ** 2).sum())

        def ac(d):
            return float(np.dot(w[:M - d], w[d:]))
        if out2 <= eps * eps * tot and ac(M // 2) <= eta * tot and ac(3 * M // 4) <= eta * tot:
            return M, kb
    return N, 0


def decide_bias(R, k0, sigma_t, s_pool, K, bias, eps, eta, classes=None):
    """fp64 mirror of the device's round-6 rule (leaf_band.hpp: band_need / band_bias_admits).  R: FFT(h) (true units), peak at k0
    in 1..N/2.  Returns (M, kb) of the first class the strict rule OR the bias-aware rule admits, else (N, 0)."""
    classes = classes or CLASSES
    Rabs = np.abs(R)
    tot = float((Rabs ** 2).sum())
    for M in classes:
        # (round 6: the forward's ring of the 2048-sample plan holds bins 0..1151, leaf_fft_wg.hpp kWgFwdBins -- windows may cross Nyquist)
        kb = int(min(max(k0 - M // 2, 1), (FWD_BINS if N == 2048 else N // 2 + 1) - M))
        inw = np.zeros(N, bool)
        inw[kb:kb + M] = True
        out2 = float((Rabs[~inw] ** 2).sum())
        # what every window drops AT DC -- bins 0, -1 .. -63 -- counts 10 times (kBandAdjacentDC / kBandAdjacent; not under round 5's rule)
        dcb = np.r_[Rabs[0:1], Rabs[N - 63:]] ** 2
        outdc, mxdc = float(dcb.sum()), float(dcb.max())
        o2 = out2 + (9.0 * outdc if eta < 2e-4 else 0.0)
        w = Rabs[kb:kb + M]
        ac_a, ac_b = float(np.dot(w[:M - M // 2], w[M // 2:])), float(np.dot(w[:M - 3 * M // 4], w[3 * M // 4:]))
        # mirrored pairs of a window across Nyquist, 2 (k - N/2) >= 5 M / 16 (leaf_band.hpp: pmir)
        pm = sum(Rabs[k] * Rabs[N - k] for k in range(max(kb, N // 2 + 5 * M // 32), kb + M) if kb <= N - k < kb + M)
        # the bias-free part: eta = kBandEtaFree = 2e-6 by default (--eta 2e-4 is round 5's rule, the strict flag)
        if o2 <= eps * eps * tot and ac_a <= eta * tot and ac_b <= eta * tot and pm <= eta * tot:
            return M, kb
        if not (ac_a <= 1e-5 * tot and ac_b <= 1e-5 * tot):
            continue
        sk = N / (2 * np.pi * sigma_t)
        dmin = min(k0 - (kb - 1), kb + M - k0) - 2 * sk
        sp = s_pool * 0.5 * (K - 1)
        if dmin >= 8:
            wd = 2 * np.pi * dmin / N
            gam = min(1.0, np.exp(-0.5 * (wd * sp) ** 2) + 2 * np.exp(-0.5 / s_pool ** 2) / (wd * 0.95 * 2.5066283 * sp))
        else:
            gam = 1.0
        rpk = Rabs[k0 % N]
        out2 = out2 + 9.0 * outdc
        if not 2 * gam * np.sqrt(out2) <= 6e-6 * rpk:
            continue
        g0 = 2.5066283 * sp
        bq = max(6.0 * float((Rabs[~inw] ** 2).max()), 60.0 * mxdc) * g0 / (2 * 5e-6)      # (kBandAdjacentDC on what is dropped at DC)
        bc = g0 * (gam * np.sqrt(out2) / rpk) ** 2 / (4 * 5e-6 ** 2)
        wl = np.pi * M / N                                                        # the pair-sum term (kBandAliasEdge, kBandAliasReg, kBandMirrorW)
        ba = (max(ac_a, ac_b) + 0.25 * pm) * (5.4e-3 / wl + 0.1 * g0 * np.exp(-0.5 * (wl * sp) ** 2)) / 1e-5
        if bias >= max(bq, bc, ba, 6.2e-5):
            return M, kb
    return N, 0


def band_pooled(x, hf, gf, hop, LS, M, kb_neg, lh, beta):
    """pooled sums of one filter through M-point inverse transforms; kb_neg = first bin of the window on the H[-k] axis"""
    K = hf.shape[0]
    T = x.shape[0]
    padl = K // 2 + K % 2 - 1
    TP = (T - 1) // hop + 1
    nblk = (T + LS - 1) // LS
    D = N // M
    Hn = np.fft.fft(hf, N)
    Hn = np.concatenate([Hn[:1], Hn[:0:-1]])
    binsel = (kb_neg + np.arange(M)) % N
    phi, lphi = kaiser_lowpass(D, lh, beta)
    Gt = D * np.convolve(gf, phi)                             # index i <-> tau = i - lphi
    q0 = lphi // D + 2
    E = np.zeros((T + D - 1) // D + 2 * q0 + N // D)
    ecirc = []
    for c in range(nblk):
        idx = c * LS - padl + np.arange(N)
        seg = np.where((idx >= 0) & (idx < T), x[np.clip(idx, 0, T - 1)], 0.0)
        z = np.fft.ifft(np.fft.fft(seg)[binsel] * Hn[binsel]) * (M / N)
        ed = np.abs(z) ** 2
        ecirc.append(ed)
        mcount = LS // D
        E[q0 + c * LS // D:q0 + c * LS // D + mcount] = ed[:mcount]
    phic = np.fft.fft(np.roll(np.concatenate([phi, np.zeros(N - len(phi))]), -lphi))
    pt = np.zeros(TP)
    edge = np.zeros(TP, bool)
    for m in range(TP):
        ws = m * hop - padl
        if ws - lphi < 0 or ws + K - 1 + lphi >= T:           # cut by the clip, or the tails of G~ would reach past it
            edge[m] = True
            acc = 0.0
            for c in range(nblk):
                lo, hi = max(c * LS, 0, ws), min((c + 1) * LS, T, ws + K)
                if lo >= hi:
                    continue
                W = np.zeros(N)
                W[(np.arange(lo, hi) - c * LS) % N] = gf[np.arange(lo, hi) - ws]
                Wt = D * np.real(np.fft.ifft(np.fft.fft(W) * phic))
                acc += np.dot(Wt[::D], ecirc[c])
            pt[m] = acc
            continue
        qlo = -((lphi - ws) // D)
        qhi = (ws + K - 1 + lphi) // D
        qs = np.arange(qlo, qhi + 1)
        pt[m] = np.dot(Gt[qs * D - ws + lphi], E[q0 + qs])
    return pt, edge


def make_signal(kind, T, rng):
    if kind == "uniform":
        return rng.uniform(-1, 1, T)
    if kind == "normal":
        return rng.standard_normal(T)
    if kind == "impulses":
        x = np.zeros(T)
        x[rng.integers(0, T, 40)] = rng.uniform(-1, 1, 40)
        return x
    if kind == "chirp":
        n = np.arange(T)
        return np.sin(np.pi * n * n / (2 * T)) + 0.001 * rng.standard_normal(T)
    if kind == "pink":
        X = np.fft.rfft(rng.standard_normal(T))
        X /= np.sqrt(np.maximum(np.arange(len(X)), 1.0))
        x = np.fft.irfft(X, T)
        return x / np.abs(x).max()
    if kind == "tones":
        n = np.arange(T)
        return sum(rng.uniform(0.1, 1) * np.sin(rng.uniform(0, np.pi) * n + rng.uniform(0, 6)) for _ in range(6)) / 6
    raise SystemExit("signal?")


def run_once(kern, pool_w, x, sr, args, verbose=True):
    F = kern.shape[0]
    K, hop = int(sr * 25.0 // 1000 + 1), int(sr * 10.0 // 1000)
    unit = int(np.lcm(hop, 64))
    LS = (N - (K - 1)) // unit * unit
    h = taps(kern, K)
    s_pool = np.clip(pool_w, 2.0 / K, 0.5)
    j = np.arange(K)
    g = np.exp(-0.5 * ((j - 0.5 * (K - 1)) / (s_pool[:, None] * 0.5 * (K - 1))) ** 2)
    p_ref = exact(x, h, g, hop)
    chosen, worst = {}, (0.0, 0.0)
    c_ = np.sqrt(2 * np.log(2)) / np.pi
    for f in range(F):
        Hn = np.fft.fft(h[f], N)
        Rabs = np.abs(Hn)                                        # FFT(h): the peak sits at k0 = mu N / 2 pi in 0..1024
        mu = float(np.clip(kern[f, 0], 0, np.pi))
        k0 = int(round(mu * N / (2 * np.pi)))
        M, kb = decide(Rabs, k0, args.eps, args.eta)
        chosen[M] = chosen.get(M, 0) + 1
        if M == N:
            if verbose:
                print(f" {f:3d} sigma {np.clip(kern[f, 1], 4 * c_, K * c_):6.2f} bin {k0:4d} -> full")
            continue
        kb_neg = (N - (kb + M - 1)) % N                           # the same M bins on the H[-k] axis, ascending
        pt, edge = band_pooled(x, h[f], g[f], hop, LS, M, kb_neg, args.lh, args.beta)
        scale = np.abs(p_ref[f]).max()
        ei = np.abs(pt - p_ref[f])[~edge].max() / scale if (~edge).any() else 0.0
        ee = np.abs(pt - p_ref[f])[edge].max() / scale if edge.any() else 0.0
        worst = (max(worst[0], ei), max(worst[1], ee))
        if verbose:
            print(f" {f:3d} sigma {np.clip(kern[f, 1], 4 * c_, K * c_):6.2f} bin {k0:4d} -> {M:4d} @ {kb:4d}  err interior {ei:.1e}  edge {ee:.1e}")
    return chosen, worst


def bias_fuzz(args, sr, K, rng):
    """Seeded cases of the round-6 rule in fp64: random (mu, sigma, pooling width, bias) per filter, signals incl. the rule's own
    worst cases (a full-scale tone on the largest dropped bin; a weak tone in the filter's core beside a strong dropped one; a tone next
    to DC), |x| <= 1.  Reported: how many filters the bias admitted beyond the strict rule and the worst error of a pooled value
    relative to bias + pooled energy (what the output sees), for the newly admitted filters."""
    hop = int(sr * 10.0 // 1000)
    unit = int(np.lcm(hop, 64))
    LS = (N - (K - 1)) // unit * unit
    c_ = np.sqrt(2 * np.log(2)) / np.pi
    worst, n_new, n_all = 0.0, 0, 0
    for it in range(args.bias_fuzz):
        F = 12
        mu = rng.uniform(0.0, np.pi, F)
        sg = rng.uniform(8, 0.3 * K, F) if it % 2 else np.exp(rng.uniform(np.log(2.0), np.log(K * c_), F))
        kern = np.stack([mu, sg], 1)
        pool_w = rng.choice([0.4, 0.4, 0.5, 0.3, 0.2, 0.1, 0.05, 0.005], F)
        bias = rng.choice([1.0, 1.0, 3.0, 0.3, 0.1, 0.02], F)
        h = taps(kern, K)
        s_pool = np.clip(pool_w, 2.0 / K, 0.5)
        j = np.arange(K)
        g = np.exp(-0.5 * ((j - 0.5 * (K - 1)) / (s_pool[:, None] * 0.5 * (K - 1))) ** 2)
        T = int(rng.integers(2 * LS + 100, 4 * LS))
        n = np.arange(T)
        picks = []
        for f in range(F):
            R = np.fft.fft(h[f], N)
            k0 = int(round(float(np.clip(mu[f], 0, np.pi)) * N / (2 * np.pi)))
            sgc = float(np.clip(sg[f], 4 * c_, K * c_))
            Ms, kbs = decide(np.abs(R), k0, args.eps, args.eta)
            Mb, kbb = decide_bias(R, k0, sgc, float(s_pool[f]), K, float(bias[f]), args.eps, args.eta)
            n_all += 1
            if Mb != N and (Ms == N or args.all_picks):          # (--all-picks: the filters the bias-free part admits as well, under their drawn bias)
                picks.append((f, Mb, kbb, R, k0))
        if not picks:
            continue
        n_new += len(picks)
        f, M, kb, R, k0 = picks[int(rng.integers(len(picks)))]
        out = np.ones(N, bool)
        out[kb:kb + M] = False
        kmax = int(np.argmax(np.where(out, np.abs(R), 0.0)))
        kmax = kmax if kmax <= N // 2 else N - kmax                     # a real tone: either sign lands on it
        kinds = {"uniform": rng.uniform(-1, 1, T),
                 "tone on the largest dropped bin": np.sin(2 * np.pi * kmax / N * n + 0.3),
                 "weak core + strong dropped": 0.02 * np.sin(2 * np.pi * k0 / N * n) + 0.98 * np.sin(2 * np.pi * kmax / N * n + 1.0),
                 "tone next to DC": np.sin(2 * np.pi * 1.5 / N * n),
                 "tone next to Nyquist": np.sin(2 * np.pi * (N / 2 - 1.5) / N * n),
                 # (late round 6: the kinds that found the pair-sum and DC-edge holes on the device)
                 "pair 0.45 M apart": 0.5 * np.sin(2 * np.pi * (k0 - 0.225 * M + 0.3) / N * n) + 0.5 * np.sin(2 * np.pi * (k0 + 0.225 * M) / N * n + 1.0),
                 "pair 0.3 M apart": 0.5 * np.sin(2 * np.pi * (k0 - 0.15 * M + 0.3) / N * n) + 0.5 * np.sin(2 * np.pi * (k0 + 0.15 * M) / N * n + 1.0),
                 "DC offset + weak core": 0.82 + 0.02 * np.sin(2 * np.pi * k0 / N * n),
                 "step in mid-clip": 0.9 * (n > T // 2) - 0.45,
                 "tone 0.2 M below Nyquist": np.sin(2 * np.pi * (N / 2 - 0.2 * M + 0.3) / N * n)}
        kind = list(kinds)[it % len(kinds)]
        x = kinds[kind]
        p_ref = exact(x, h[f:f + 1], g[f:f + 1], hop)[0]
        kb_neg = (N - (kb + M - 1)) % N
        pt, edge = band_pooled(x, h[f], g[f], hop, LS, M, kb_neg, args.lh, args.beta)
        err = float((np.abs(pt - p_ref) / (bias[f] + p_ref)).max())
        worst = max(worst, err)
        print(f"case {it:3d} {kind:32s} filter: bin {k0:4d} sigma {np.clip(sg[f], 4 * c_, K * c_):6.1f} pool_w {pool_w[f]:.3f} bias {bias[f]:.2f} -> {M:4d} @ {kb:4d}: "
              f"err / (bias + p) {err:.2e}   (pooled energy up to {p_ref.max():.2e})", flush=True)
    print(f"{n_new} of {n_all} filters admitted by the bias beyond the strict rule; worst error relative to bias + pooled energy {worst:.2e}")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--sr", type=int, default=16000)
    ap.add_argument("--filters", type=int, default=40)
    ap.add_argument("--T", type=int, default=8000)
    ap.add_argument("--lh", type=int, default=12)
    ap.add_argument("--beta", type=float, default=12.0)
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--eps", type=float, default=3e-6)
    ap.add_argument("--eta", type=float, default=1e-4)
    ap.add_argument("--all-picks", action="store_true")
    ap.add_argument("--pool-w", type=float, default=0.4)
    ap.add_argument("--signal", default="uniform")
    ap.add_argument("--fuzz", type=int, default=0)
    ap.add_argument("--N", type=int, default=2048, help="block length: 2048, or 4096 (the 32 kHz plan; with --classes 512)")
    ap.add_argument("--classes", default="", help="comma-separated transform lengths to try (default 256,512)")
    ap.add_argument("--bias-fuzz", type=int, default=0, help="round 6: N cases of the bias-aware rule (decide_bias) incl. adversarial tones; "
                                                            "errors relative to bias + pooled energy")
    args = ap.parse_args()
    global N, CLASSES
    N = args.N
    if args.classes:
        CLASSES = tuple(int(c) for c in args.classes.split(","))
    sr, F = args.sr, args.filters
    K = int(sr * 25.0 // 1000 + 1)
    rng = np.random.default_rng(args.seed)
    if args.bias_fuzz:
        return bias_fuzz(args, sr, K, rng)
    if not args.fuzz:
        x = make_signal(args.signal, args.T, rng)
        chosen, worst = run_once(default_kernel(F, sr, K), np.full(F, args.pool_w), x, sr, args)
        work = sum(n * (m * np.log2(m)) / (N * np.log2(N)) for m, n in chosen.items())
        print(f"classes {dict(sorted(chosen.items()))}: transform work {work:.1f} of {F}; worst error interior {worst[0]:.1e} edge {worst[1]:.1e}")
        return
    c_ = np.sqrt(2 * np.log(2)) / np.pi
    tot, W = {}, (0.0, 0.0)
    for it in range(args.fuzz):
        F = 16
        mu = rng.uniform(-0.2, np.pi + 0.2, F)
        sg = np.exp(rng.uniform(np.log(1.0), np.log(1.3 * K * c_), F))
        if it % 3 == 0:                                          # sigmas at the class boundaries of typical filters
            sg = rng.uniform(8, 60, F)
        kern = np.stack([mu, sg], 1)
        pool_w = rng.uniform(0.0, 0.7, F)
        kind = ("uniform", "normal", "impulses", "chirp", "pink", "tones")[it % 6]
        x = make_signal(kind, rng.integers(1700, 6000), rng)
        chosen, worst = run_once(kern, pool_w, x, sr, args, verbose=False)
        for k, v in chosen.items():
            tot[k] = tot.get(k, 0) + v
        W = (max(W[0], worst[0]), max(W[1], worst[1]))
        print(f"fuzz {it:3d} {kind:9s} T {x.shape[0]:5d} classes {dict(sorted(chosen.items()))} worst interior {worst[0]:.1e} edge {worst[1]:.1e}")
    print(f"total classes {dict(sorted(tot.items()))}; worst interior {W[0]:.1e} edge {W[1]:.1e}")


if __name__ == "__main__":
    main()
