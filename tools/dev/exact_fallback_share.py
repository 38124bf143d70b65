"""Share of first-pass windows that precision="exact" sends to the float64 transform, on the golden fixtures' frames and on
synthetic frames of several kinds (development aid)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch

from torchpiv_amd import engine, synth


def share(a, b, ws, ov, what):
    A = torch.from_numpy(np.ascontiguousarray(a))[None].cuda()
    B = torch.from_numpy(np.ascontiguousarray(b))[None].cuda()
    H, W = a.shape
    plan = engine.Plan(H, W, ws, ov, n_pass=1, max_batch=1, precision="exact")
    plan.run(A, B)
    n_fb = plan.exact_fallbacks()
    n = plan.geometry[0][2] * plan.geometry[0][3]
    print(f"{what:40s} ws {ws:3d} ov {ov:3d} {H}x{W}: {n_fb:6d} of {n:6d} windows ({100.0 * n_fb / n:6.2f} %)")
    plan.close()


if __name__ == "__main__":
    root = os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests", "golden")
    for f in sorted(os.listdir(root)):
        if not f.endswith(".npz"):
            continue
        g = np.load(os.path.join(root, f), allow_pickle=True)
        for k in g.files:
            if k.endswith("_a") and k[:-2] + "_b" in g.files and g[k].ndim == 2 and g[k].dtype == np.uint8:
                a, b = g[k], g[k[:-2] + "_b"]
                for ws in (32, 64, 128):
                    if ws <= min(a.shape):
                        share(a, b, ws, ws // 2, f"{f[:-4]}:{k[:-2]}")
    for kind in ("wavy", "vortex", "shear"):
        for noise in (0.0, 3.0):
            a, b = synth.make_pair(1024, 1024, 77, kind=kind, noise=noise)
            share(a.numpy(), b.numpy(), 64, 32, f"synth {kind} noise {noise:g}")
