"""How far is the float32 locating pass of precision="exact" from the exact map?  (GPU; prints a table.)

For families of 64x64 windows -- particle images at several noise levels, pure noise, and windows built to make the float32
transform look bad (nearly orthogonal patterns, one bright pixel on a pedestal, two grey levels, a saturated frame with a
few dark pixels) -- the float32 correlation maps of the tile kernel (debug hook) are compared with the float64 maps:

  err / R   largest cell error relative to the map range
  err / E   ... relative to E = |a - mean a| |b - mean b| / (mean a mean b) <= E+, the scale the transform's rounding follows
            (the band of the locating pass is 2 Gamma (1 + 1/16) E+ with the PROVEN Gamma = 247 u = 1.47e-5)
  R / E     contrast of the map (round 4 sent windows below 0.028 to the float64 transform; the proven band needs no such guard)

and the fields of precision="exact" with those of precision="f64" (any difference above 1e-9 px is a wrong decision that
went unnoticed).
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch

from torchpiv_amd import engine, synth


def families(n=64, seed=0):
    rng = np.random.default_rng(seed)
    W = 64
    yy, xx = np.mgrid[0:W, 0:W]
    fam = {}
    for noise in (0.0, 2.0, 8.0, 24.0):
        A, B = synth.make_batch(1, 1024, 1024, device="cpu", noise=noise, first_index=int(noise) + 3)
        a = A[0].numpy().reshape(16, 64, 16, 64).transpose(0, 2, 1, 3).reshape(-1, 64, 64)[:n]
        b = B[0].numpy().reshape(16, 64, 16, 64).transpose(0, 2, 1, 3).reshape(-1, 64, 64)[:n]
        fam[f"particles noise {noise:g}"] = (a, b)
    fam["uniform noise"] = (rng.integers(0, 256, (n, W, W)), rng.integers(0, 256, (n, W, W)))
    fam["low-contrast noise 100..103"] = (rng.integers(100, 104, (n, W, W)), rng.integers(100, 104, (n, W, W)))
    fam["two levels 200/201"] = (200 + rng.integers(0, 2, (n, W, W)), 200 + rng.integers(0, 2, (n, W, W)))
    k = rng.integers(1, 31, (n, 1, 1))
    fam["orthogonal sinusoids"] = (np.rint(127 + 100 * np.sin(2 * np.pi * k * xx[None] / W)),
                                   np.rint(127 + 100 * np.sin(2 * np.pi * k * yy[None] / W)))
    fam["same sinusoid, other phase"] = (np.rint(127 + 100 * np.sin(2 * np.pi * k * xx[None] / W)),
                                         np.rint(127 + 100 * np.sin(2 * np.pi * k * xx[None] / W + 1.0)))
    one = np.full((n, W, W), 10)
    one[np.arange(n), rng.integers(0, W, n), rng.integers(0, W, n)] = 255
    two = np.full((n, W, W), 10)
    two[np.arange(n), rng.integers(0, W, n), rng.integers(0, W, n)] = 255
    fam["one bright pixel on a pedestal"] = (one, two)
    sat = np.full((n, W, W), 255)
    sat2 = np.full((n, W, W), 255)
    for i in range(n):
        sat[i, rng.integers(0, W, 5), rng.integers(0, W, 5)] = 0
        sat2[i, rng.integers(0, W, 5), rng.integers(0, W, 5)] = 0
    fam["saturated, five dark pixels"] = (sat, sat2)
    fam["checkerboard vs stripes"] = (np.broadcast_to(255 * ((xx + yy) & 1), (n, W, W)) + rng.integers(0, 2, (n, W, W)) * 0,
                                      np.broadcast_to(255 * (xx & 1), (n, W, W)) + rng.integers(0, 3, (n, W, W)))
    fam["ramp vs noise"] = (np.broadcast_to(xx * 4, (n, W, W)), rng.integers(0, 256, (n, W, W)))
    # background-subtracted recordings: 3x3 particle images on a TRUE zero, most map cells exactly 0
    za, zb = np.zeros((n, W + 8, W + 8)), None
    for i in range(n):
        for _ in range(14):
            y, x = rng.integers(3, W + 4), rng.integers(3, W + 4)
            za[i, y - 1:y + 2, x - 1:x + 2] += rng.uniform(80, 200)
    fam["particles on a true-zero background"] = (za[:, 2:2 + W, 2:2 + W], za[:, 4:4 + W, 1:1 + W])
    return {k_: (np.clip(a, 0, 255).astype(np.uint8), np.clip(b, 0, 255).astype(np.uint8)) for k_, (a, b) in fam.items()}


def main():
    print(f"{'family':34s} {'err/R':>10s} {'err/E':>10s} {'min R/E':>9s} {'float64 path':>12s} {'max |exact - f64| px':>20s} {'masks':>6s}")
    for name, (a, b) in families().items():
        n = a.shape[0]
        A = torch.from_numpy(np.ascontiguousarray(a)).cuda()
        B = torch.from_numpy(np.ascontiguousarray(b)).cuda()
        _, _, _, _, corr = engine.debug_pass(0, A, B, 64, 0, precision="fast")
        c32 = corr.cpu().numpy().reshape(n, 64, 64).astype(np.float64)
        af, bf = a.astype(np.float64), b.astype(np.float64)
        ma, mb = af.mean(axis=(1, 2), keepdims=True), bf.mean(axis=(1, 2), keepdims=True)
        c64 = np.fft.fftshift(np.fft.irfft2(np.conj(np.fft.rfft2(af / ma)) * np.fft.rfft2(bf / mb), s=(64, 64)), axes=(1, 2))
        c64 = c64 - c64.min(axis=(1, 2), keepdims=True) + 1e-7
        err = np.abs(c32 - c64).max(axis=(1, 2))
        R = c64.max(axis=(1, 2)) - c64.min(axis=(1, 2))
        E = np.sqrt(((af - ma) ** 2).sum(axis=(1, 2)) * ((bf - mb) ** 2).sum(axis=(1, 2))) / (ma * mb).reshape(-1)
        ok = (R > 0) & (E > 0)
        plan = engine.Plan(64, 64, 64, 0, n_pass=1, max_batch=n, precision="exact")
        ue, ve, ie = plan.run(A, B)
        n_fb = plan.exact_fallbacks()
        uf, vf, i_f = engine.pass1(A, B, 64, 0, precision="f64")
        d = max(float((ue - uf).abs().max()), float((ve - vf).abs().max()))
        with np.errstate(all="ignore"):
            print(f"{name:34s} {np.nanmax(np.where(ok, err / R, np.nan)):10.2e} {np.nanmax(np.where(ok, err / E, np.nan)):10.2e} "
                  f"{np.nanmin(np.where(ok, R / E, np.nan)):9.2e} {n_fb:7d}/{n:<4d} {d:20.2e} {int((ie != i_f).sum()):6d}")


if __name__ == "__main__":
    main()
