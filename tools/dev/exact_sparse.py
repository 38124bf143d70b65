"""Share of undecided windows of precision="exact" on very sparse particle images with a true-zero background (1 ... 8 particle
images per 64x64 window): with ONE particle pair the map outside the peak is flat zero, the second-peak band overflows and every
window takes the float64 transform; from two on it is 0.4 ... 3 % (development aid)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from torchpiv_amd import engine
rng = np.random.default_rng(3)
W, n = 64, 256
for npart in (1, 2, 3, 5, 8):
    za = np.zeros((n, W + 8, W + 8))
    for i in range(n):
        for _ in range(npart):
            y, x = rng.integers(8, W - 4), rng.integers(8, W - 4)
            za[i, y - 1:y + 2, x - 1:x + 2] += rng.uniform(80, 200)
    a = np.clip(za[:, 2:2 + W, 2:2 + W], 0, 255).astype(np.uint8)
    b = np.clip(za[:, 4:4 + W, 1:1 + W], 0, 255).astype(np.uint8)
    A, B = torch.from_numpy(a).cuda(), torch.from_numpy(b).cuda()
    plan = engine.Plan(64, 64, 64, 0, n_pass=1, max_batch=n, precision="exact")
    ue, ve, ie = plan.run(A, B)
    nfb = plan.exact_fallbacks()
    uf, vf, i_f = engine.pass1(A, B, 64, 0, precision="f64")
    d = max(float((ue - uf).abs().max()), float((ve - vf).abs().max()))
    print(f"{npart} particles per window, true-zero background: undecided {nfb}/{n}, max |exact - f64| {d:.1e}, flags differing {int((ie != i_f).sum())}")
