"""precision="exact" at every first-pass window size (development aid; GPU): fields against the float64 kernels, share of
windows that took the float64 transform, pass-1 time at the three precisions."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch

from torchpiv_amd import engine, synth


def run(ws, ov, size, batch, noise=2.0):
    A, B = synth.make_batch(batch, size, size, device="cuda", noise=noise, first_index=ws + 3)
    pe = engine.Plan(size, size, ws, ov, n_pass=1, max_batch=batch, precision="exact")
    name = pe.kernel_name(0)
    ue, ve, ie = pe.run(A, B)
    torch.cuda.synchronize()
    n_fb = pe.exact_fallbacks()
    n_win = batch * pe.geometry[0][2] * pe.geometry[0][3]
    uf, vf, i_f = engine.pass1(A, B, ws, ov, precision="f64")
    d = max(float((ue - uf).abs().max()), float((ve - vf).abs().max()))
    nm = int((ie != i_f).sum())
    times = {}
    for prec in ("exact", "f64", "fast"):
        pl = pe if prec == "exact" else engine.Plan(size, size, ws, ov, n_pass=1, max_batch=batch, precision=prec)
        pl.run(A, B)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(3):
            pl.run(A, B)
        torch.cuda.synchronize()
        times[prec] = (time.perf_counter() - t0) / 3 / batch * 1e6
        if pl is not pe:
            pl.close()
    pe.close()
    print(f"ws {ws:3d} ov {ov:3d} {size}^2 x{batch}: {name:46s} fallback {n_fb:6d}/{n_win:<8d} ({100.0 * n_fb / n_win:5.2f} %)  "
          f"max|exact-f64| {d:.1e} px  masks differing {nm}  us/pair exact {times['exact']:.1f} f64 {times['f64']:.1f} fast {times['fast']:.1f}",
          flush=True)
    return d, nm


if __name__ == "__main__":
    sizes = [int(a) for a in sys.argv[1:]] or [8, 16, 32, 64, 128, 12, 24, 28, 40, 42, 48, 56, 96, 22, 100, 126, 10, 6]
    bad = 0
    for ws in sizes:
        size = 1024 if ws >= 24 else 512
        d, nm = run(ws, ws // 2, size, 4)
        bad += (d > 1e-11) + (nm > 0)
    sys.exit(1 if bad else 0)
