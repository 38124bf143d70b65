"""CPU model of the "exact" first pass (precision="exact") against the oracle.

The scheme: a float32 FFT map only LOCATES cells -- the arg-max, the cells that can be the second peak, the cells that
can be the minimum -- with an error band around every decision; the VALUES the sub-pixel fit and the validity ratio use
are exact integer correlation sums  S(d) = sum_p a[p] b[p + d]  of the uint8 windows (what the reference's float64 FFT
approximates to ~1e-16 relative), turned into the reference's map value  (S - S_min) n^4 / (sum a sum b) + 1e-7.  A window
whose float32 map leaves a decision open inside the band goes to the float64 FFT kernel instead.

This file is the numpy statement of that scheme (the device code in csrc/xcorr_exact.hip follows it step by step and
is compared with it cell for cell in tests/test_gpu_exact.py); here it is checked against the oracle's float64 pass 1.
"""
import numpy as np
import pytest
import torch

from oracle import piv_oracle as O

ETA = 6.66              # per radix-2 level of a float32 FFT: twiddle error + 4 roundings (piv_kernels.h: EXACT_ETA)
U32 = 2.0 ** -24        # unit roundoff of float32
MAX_SECOND = 3          # candidates carried per window (more -> float64 fallback)
MAX_MIN = 4


def band_coef(W):
    """2 Gamma(W) (1 + 1/16): the proven bound on the float32 map's cell error, per unit of E+ (piv_kernels.h: exact_band_coef)."""
    levels = 2 * int(np.ceil(np.log2(W)))
    return 2.0 * (3 * levels * ETA + 7) * (1 + 1 / 16) * U32


def e_plus(a, b):
    """E+ = (|a'|^2 + |b'|^2) / 2 with a' = a / mean(a) - 1: the scale of the float32 transform's rounding error."""
    af, bf = a.astype(np.float64), b.astype(np.float64)
    return 0.5 * (((af / af.mean() - 1) ** 2).sum() + ((bf / bf.mean() - 1) ** 2).sum())


def excluded(q, m, wv, W):
    """B:346-358: q is zeroed iff q == clip(m + i + W j) for some |i|, |j| <= wv."""
    KD = W * W
    for j in range(-wv, wv + 1):
        for i in range(-wv, wv + 1):
            if min(max(m + i + W * j, 0), KD - 1) == q:
                return True
    return False


def f32_map(a, b):
    """What the float32 tile kernel computes: mean-removed, mean-normalised windows through a float32 FFT, fftshift."""
    af = a.astype(np.float32)
    bf = b.astype(np.float32)
    ma, mb = af.mean(dtype=np.float32), bf.mean(dtype=np.float32)
    ta = torch.from_numpy((af - ma) / ma)
    tb = torch.from_numpy((bf - mb) / mb)
    c = torch.fft.fftshift(torch.fft.irfft2(torch.fft.rfft2(ta).conj() * torch.fft.rfft2(tb)))
    return c.numpy()


def exact_sum(a, b, q):
    W = a.shape[0]
    dy, dx = q // W - W // 2, q % W - W // 2
    return int((a.astype(np.int64) * np.roll(b.astype(np.int64), (-dy, -dx), axis=(0, 1))).sum())


def candidates(cmap, wv, band):
    """-> (m, second candidates, minimum candidates) as flat indices, or None when a decision is open."""
    W = cmap.shape[0]
    flat = cmap.reshape(-1)
    lo, hi = float(flat.min()), float(flat.max())
    band = np.float32(band)
    if not band > 0 or not hi > lo:
        return None
    top = np.flatnonzero(flat >= np.float32(hi) - band)
    if top.size != 1:
        return None
    m = int(top[0])
    ok = np.array([not excluded(q, m, wv, W) for q in range(W * W)]) if wv * 2 + 1 < W else np.zeros(W * W, bool)
    if ok.any():
        s_hi = flat[ok].max()
        second = np.flatnonzero(ok & (flat >= s_hi - band))
        if second.size > MAX_SECOND:
            return None
    else:
        second = np.zeros(0, np.int64)
    mins = np.flatnonzero(flat <= np.float32(lo) + band)
    overflow = mins.size > MAX_MIN          # (true-zero backgrounds: hundreds of cells at S = 0; see exact_window)
    if overflow:
        mins = mins[:MAX_MIN - 1]
    return m, [int(s) for s in second], [int(s) for s in mins], overflow


def exact_window(a, b, wv=3, val_ratio=1.2):
    """-> (u, v, invalid) of one window by the exact scheme, or None (float64 fallback)."""
    W = a.shape[0]
    KD = W * W
    sa, sb = int(a.sum(dtype=np.int64)), int(b.sum(dtype=np.int64))
    if sa == 0 or sb == 0:
        return 0.0, 0.0, False                      # zero-mean window: NaN map in the reference -> (0, 0), valid
    cand = candidates(f32_map(a, b), wv, band_coef(W) * e_plus(a, b))
    if cand is None:
        return None
    m, second, mins, min_overflow = cand
    left, right, top, bot = m + 1, m - 1, m + W, m - W
    left = m if left >= KD - 1 else left
    right = m if right <= 0 else right
    top = m if top >= KD - 1 else top
    bot = m if bot <= 0 else bot
    S = {q: exact_sum(a, b, q) for q in {m, left, right, top, bot, *second, *mins}}
    if max(S[left], S[right], S[top], S[bot]) > S[m]:
        return None                                 # the float32 arg-max was not the exact one: the band was too narrow
    smin = min(S[q] for q in mins)
    if min(S.values()) < smin:
        return None
    if min_overflow and smin != 0:
        return None                                 # S >= 0 everywhere: only an evaluated 0 is certainly the minimum
    scale = float(W) ** 4 / (float(sa) * float(sb))
    val = lambda q: (S[q] - smin) * scale + 1e-7
    cm, cl, cr, ct, cb = val(m), val(left), val(right), val(top), val(bot)
    # (nothing outside the exclusion zone: the reference zeroes its float64 map in place and divides by the 0 it finds there)
    c2 = max(val(q) for q in second) if second else 0.0
    with np.errstate(all="ignore"):
        lm, ll, lr, lt, lb = np.log(cm), np.log(cl), np.log(cr), np.log(ct), np.log(cb)
        u = m % W + (lr - ll) / (2 * (ll + lr) - 4 * lm) - W // 2
        v = m // W + (lb - lt) / (2 * (lb + lt) - 4 * lm) - W // 2
        invalid = bool(np.float64(cm) / np.float64(c2) < val_ratio)
    return float(np.nan_to_num(u)), float(np.nan_to_num(v)), invalid


def synthetic_pair(H, W, seed, shift=(2.3, -1.6), n=None, noise=2.0):
    rng = np.random.default_rng(seed)
    n = n or H * W // 40
    py, px = rng.uniform(0, H, n), rng.uniform(0, W, n)
    amp = rng.uniform(80, 250, n)
    yy, xx = np.mgrid[0:H, 0:W]

    def render(oy, ox):
        img = np.zeros((H, W))
        for y, x, a in zip(py + oy, px + ox, amp):
            y0, x0 = int(round(y)), int(round(x))
            ys = slice(max(y0 - 3, 0), min(y0 + 4, H))
            xs = slice(max(x0 - 3, 0), min(x0 + 4, W))
            img[ys, xs] += a * np.exp(-((yy[ys, xs] - y) ** 2 + (xx[ys, xs] - x) ** 2) / 2.0)
        img += rng.normal(8, noise, img.shape)
        return np.clip(np.rint(img), 0, 255).astype(np.uint8)

    return render(0, 0), render(*shift)


@pytest.mark.parametrize("seed,shift", [(1, (2.3, -1.6)), (2, (0.0, 0.0)), (3, (-7.4, 11.2))])
def test_exact_scheme_matches_the_float64_oracle(seed, shift):
    A, B = synthetic_pair(256, 256, seed, shift)
    ws, ov = 64, 32
    u0, v0, _, _, mask0 = O.pass1(A, B, ws, ov, validate=True)
    aw, bw = O.windows(A, ws, ov), O.windows(B, ws, ov)
    n_fallback = 0
    for i, (a, b) in enumerate(zip(aw, bw)):
        r = exact_window(a, b)
        if r is None:
            n_fallback += 1
            continue
        u, v, inv = r
        # the exact sums against the reference's float64 FFT: rounding of a 64x64 float64 transform, amplified by the fit
        assert abs(u - u0.reshape(-1)[i]) < 1e-10 and abs(v - v0.reshape(-1)[i]) < 1e-10, (i, u, v)
        assert inv == bool(mask0.reshape(-1)[i]), i
    assert n_fallback <= max(1, len(aw) // 10), n_fallback


def test_exact_scheme_dead_and_identical_windows():
    rng = np.random.default_rng(5)
    a = rng.integers(0, 255, (64, 64), dtype=np.uint8)
    assert exact_window(np.zeros_like(a), a) == (0.0, 0.0, False)
    r = exact_window(a, a.copy())
    # an autocorrelation is even: with exact sums the fit is exactly zero (the property finalize_kernel enforces for FFT maps)
    assert r is not None and r[0] == 0.0 and r[1] == 0.0


def test_exact_sums_are_the_circular_correlation():
    rng = np.random.default_rng(9)
    a = rng.integers(0, 255, (16, 16), dtype=np.uint8)
    b = rng.integers(0, 255, (16, 16), dtype=np.uint8)
    c = O.xcorr_fft(a.astype(np.float64), b.astype(np.float64))
    for q in (0, 17, 100, 255, 136):
        assert abs(exact_sum(a, b, q) - c.reshape(-1)[q]) < 1e-6



def test_exact_scheme_on_a_true_zero_background():
    """Background-subtracted recordings: most map cells are exactly 0 (no particle pair overlaps at that shift) -- far more
    minimum candidates than the record holds, and still decidable: S >= 0, so the evaluated zeros are the minimum."""
    rng = np.random.default_rng(12)
    H = W = 192
    img = np.zeros((H + 8, W + 8))
    for _ in range(120):
        y, x = rng.integers(3, H + 4), rng.integers(3, W + 4)
        img[y - 1:y + 2, x - 1:x + 2] += rng.uniform(80, 200)
    A = np.clip(img[2:2 + H, 2:2 + W], 0, 255).astype(np.uint8)
    B = np.clip(img[4:4 + H, 1:1 + W], 0, 255).astype(np.uint8)
    u0, v0, _, _, m0 = O.pass1(A, B, 64, 32, validate=True)
    aw, bw = O.windows(A, 64, 32), O.windows(B, 64, 32)
    decided = 0
    for i, (a, b) in enumerate(zip(aw, bw)):
        r = exact_window(a, b)
        if r is None:
            continue
        decided += 1
        assert abs(r[0] - u0.reshape(-1)[i]) < 1e-10 and abs(r[1] - v0.reshape(-1)[i]) < 1e-10 and r[2] == bool(m0.reshape(-1)[i])
    assert decided >= len(aw) // 2, (decided, len(aw))


def test_float32_map_error_stays_inside_the_proven_bound():
    """|map32 - map| <= Gamma E+ (piv_kernels.h "The band") on the adversarial windows of
    tools/research/exact_adversarial.py (tests/golden/g12_adversarial.npz -- hill-climbed on the GPU kernel's own error) and on
    random ones, for this file's float32 FFT (pocketfft / MKL: the analysis does not depend on the radix schedule)."""
    import os
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "g12_adversarial.npz"))
    gamma = band_coef(64) / (2 * (1 + 1 / 16))
    assert abs(gamma - 246.76 * U32) < 1e-12
    rng = np.random.default_rng(7)
    fams = [g[f"w{i}"] for i in range(len(g["names"]))] + [rng.integers(0, 256, (4, 2, 64, 64)).astype(np.uint8)]
    worst = 0.0
    for P in fams:
        for a, b in P:
            if a.sum() == 0 or b.sum() == 0:
                continue
            af, bf = a.astype(np.float64), b.astype(np.float64)
            c64 = np.fft.fftshift(np.fft.irfft2(np.conj(np.fft.rfft2(af / af.mean() - 1)) * np.fft.rfft2(bf / bf.mean() - 1), s=a.shape))
            e = f32_map(a, b).astype(np.float64) - c64
            worst = max(worst, 0.5 * (e.max() - e.min()) / e_plus(a, b))
    assert 0 < worst < gamma / 8, (worst, gamma)
