"""Soak of precision="exact" against the float64 FFT path: first-pass fields of many 2048 x 2048 synthetic pairs (three
flow kinds, four noise levels, 64/32 windows), every window compared.  Prints one line per block and a total."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch

from torchpiv_amd import engine, synth

if __name__ == "__main__":
    n_blocks = int(sys.argv[1]) if len(sys.argv) > 1 else 16
    batch, H = 64, 2048
    pe = engine.Plan(H, H, 64, 32, n_pass=1, max_batch=batch, precision="exact")
    tot = dict(windows=0, undecided=0, identical=0, flags=0)
    worst = 0.0
    for blk in range(n_blocks):
        kind, noise = ("wavy", "vortex", "shear")[blk % 3], (0.0, 1.0, 4.0, 12.0)[blk % 4]
        A, B = synth.make_batch(batch, H, H, first_index=10_000 + blk * batch, kind=kind, noise=noise, device="cuda")
        ue, ve, ie = pe.run(A, B)
        n_fb = pe.exact_fallbacks()
        uf, vf, i_f = engine.pass1(A, B, 64, 32, precision="f64")
        d = torch.maximum((ue - uf).abs(), (ve - vf).abs())
        tot["windows"] += d.numel()
        tot["undecided"] += n_fb
        tot["identical"] += int((d == 0).sum())
        tot["flags"] += int((ie != i_f).sum())
        worst = max(worst, float(d.max()))
        print(f"block {blk:3d} {kind:6s} noise {noise:4.1f}: {d.numel()} windows, undecided {n_fb}, bit-identical {int((d == 0).sum())}, "
              f"max |d| {float(d.max()):.2e} px, flags differing {int((ie != i_f).sum())}", flush=True)
    print(f"TOTAL {tot['windows']} windows of {n_blocks * batch} pairs: undecided {tot['undecided']} "
          f"({100.0 * tot['undecided'] / tot['windows']:.3f} %), bit-identical {tot['identical']} "
          f"({100.0 * tot['identical'] / tot['windows']:.4f} %), max |exact - f64| {worst:.2e} px, validity flags differing {tot['flags']}")
