import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from oracle import piv_oracle as O
from torchpiv_amd import engine
g = np.load("tests/golden/g5_generator.npz")
A, B = g["frames_a"], g["frames_b"]
for run in ("r2", "r3", "r4"):
    ws, ov, mp_, mode, dt = (int(t) for t in g[run + "_kw"])
    mode = ("DWS", "CWS")[mode]
    for k in range(4):
        a, b = A[k], B[k]
        plan = engine.Plan(a.shape[0], a.shape[1], ws, ov, n_pass=mp_, mode=mode, max_batch=1)
        u, v, inv = plan.run(torch.from_numpy(a).cuda(), torch.from_numpy(b).cuda())
        ou, ov_, x, y, oval = O.pass1(a, b, ws, ov, validate=True)
        w, o = ws, ov
        for p in range(mp_):
            if p > 0:
                w, o = w // 2, o // 2
                it = O.ITER[mode](a.shape, w, o)
                ou, ov_, x, y, oval, odu, odv, ou0, ov0, ou2, ov2 = it(a, b, x, y, ou, ov_, oval, debug=True)
            if p < mp_ - 1:
                gu, gv, gi = plan.pass_fields(p, 1)
            else:
                gu, gv, gi = u, v, inv
            gu, gv, gi = gu[0].cpu().numpy(), gv[0].cpu().numpy(), gi[0].cpu().numpy().astype(bool)
            flips = gi != oval
            err = np.maximum(np.abs(gu - ou), np.abs(gv - ov_))
            bad = (err > 1e-3) & ~flips
            print(f"{run} {mode} pair {k} pass {p} ws {w}: n {gu.size} flips {flips.sum()} bad {bad.sum()} "
                  f"invalid(ref) {oval.sum()} max_err_ok {err[~bad & ~flips].max():.2e}")
            if bad.any() and p == 0:
                idx = np.argwhere(bad)[:3]
                for (r, c) in idx:
                    print("   ", r, c, gu[r, c], ou[r, c], gv[r, c], ov_[r, c], gi[r, c], oval[r, c])
        plan.close()
