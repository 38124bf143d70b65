"""Soak of precision="exact" against the float64 FFT path: first-pass fields of many synthetic pairs (three flow kinds, four
noise levels), every window compared -- 2048 x 2048 frames at 64/32 (the headline geometry), then the other window sizes
(8 ... 128 and generic ones) on smaller frames.  Also prints the distribution of the map contrast R / E+ (R = max - min of the
normalised map, E+ the scale of the proven error bound) over a sample of the windows: the band of the locating
pass is 3.1e-5 E+ whatever the contrast, i.e. 3.1e-5 / (R / E+) of the map's range.

    python tools/research/exact_soak.py [blocks of 64 pairs at 2048^2, default 16]
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch

from torchpiv_amd import engine, synth


def contrast(A, B, ws, ov, n_pairs=2):
    """R / E+ of every window of the first n_pairs pairs (float64 maps on the device)."""
    out = []
    st = ws - ov
    for k in range(min(n_pairs, A.shape[0])):
        a = A[k].double().unfold(0, ws, st).unfold(1, ws, st).reshape(-1, ws, ws)
        b = B[k].double().unfold(0, ws, st).unfold(1, ws, st).reshape(-1, ws, ws)
        ma, mb = a.mean(dim=(1, 2), keepdim=True), b.mean(dim=(1, 2), keepdim=True)
        ok = (ma.reshape(-1) > 0) & (mb.reshape(-1) > 0)
        an, bn = a[ok] / ma[ok] - 1, b[ok] / mb[ok] - 1
        c = torch.fft.irfft2(torch.fft.rfft2(an).conj() * torch.fft.rfft2(bn), s=(ws, ws))
        R = c.amax(dim=(1, 2)) - c.amin(dim=(1, 2))
        ep = 0.5 * ((an ** 2).sum(dim=(1, 2)) + (bn ** 2).sum(dim=(1, 2)))
        out.append((R / ep)[ep > 0])
    return torch.cat(out)


def soak(ws, ov, H, batch, n_blocks, tag, first=10_000):
    pe = engine.Plan(H, H, ws, ov, n_pass=1, max_batch=batch, precision="exact")
    tot = dict(windows=0, undecided=0, identical=0, flags=0)
    worst = 0.0
    ratios = []
    for blk in range(n_blocks):
        kind, noise = ("wavy", "vortex", "shear")[blk % 3], (0.0, 1.0, 4.0, 12.0)[blk % 4]
        A, B = synth.make_batch(batch, H, H, first_index=first + blk * batch, kind=kind, noise=noise, device="cuda")
        ue, ve, ie = pe.run(A, B)
        n_fb = pe.exact_fallbacks()
        uf, vf, i_f = engine.pass1(A, B, ws, ov, precision="f64")
        d = torch.maximum((ue - uf).abs(), (ve - vf).abs())
        tot["windows"] += d.numel()
        tot["undecided"] += n_fb
        tot["identical"] += int((d == 0).sum())
        tot["flags"] += int((ie != i_f).sum())
        worst = max(worst, float(d.max()))
        ratios.append(contrast(A, B, ws, ov, 1))
        if tag == "64/32":
            print(f"block {blk:3d} {kind:6s} noise {noise:4.1f}: {d.numel()} windows, undecided {n_fb}, bit-identical {int((d == 0).sum())}, "
                  f"max |d| {float(d.max()):.2e} px, flags differing {int((ie != i_f).sum())}", flush=True)
    pe.close()
    r = torch.cat(ratios)
    q = torch.quantile(r, torch.tensor([0.001, 0.01, 0.1, 0.5, 0.9, 0.99], dtype=r.dtype, device=r.device)).tolist()
    print(f"TOTAL {tag:>7s} ({H}^2): {tot['windows']} windows of {n_blocks * batch} pairs: undecided {tot['undecided']} "
          f"({100.0 * tot['undecided'] / tot['windows']:.3f} %), bit-identical {tot['identical']} "
          f"({100.0 * tot['identical'] / tot['windows']:.4f} %), max |exact - f64| {worst:.2e} px, validity flags differing {tot['flags']}; "
          f"R/E+ over {r.numel()} sampled windows: min {float(r.min()):.3f}, 0.1 % {q[0]:.3f}, 1 % {q[1]:.3f}, 10 % {q[2]:.3f}, "
          f"median {q[3]:.3f}, 90 % {q[4]:.3f}, 99 % {q[5]:.3f}", flush=True)


if __name__ == "__main__":
    n_blocks = int(sys.argv[1]) if len(sys.argv) > 1 else 16
    soak(64, 32, 2048, 64, n_blocks, "64/32")
    for ws, H, batch in ((32, 1024, 32), (128, 2048, 16), (16, 1024, 8), (8, 512, 8), (24, 1024, 8), (42, 1024, 8), (48, 1024, 8), (22, 512, 8)):
        soak(ws, ws // 2, H, batch, 8, f"{ws}/{ws // 2}", first=50_000 + ws * 100)
