import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
import numpy as np, torch
from oracle import piv_oracle as O
from torchpiv_amd import engine
from test_gpu_parity import fp32_noise_excuse, constant_windows
g = np.load("tests/golden/g5_generator.npz")
a, b = g["frames_a"][0], g["frames_b"][0]
H, W = a.shape
for prec in ("fast", "reference"):
    plan = engine.Plan(H, W, 32, 16, n_pass=3, mode="CWS", max_batch=1, precision=prec)
    u, v, inv = plan.run(torch.from_numpy(a).cuda(), torch.from_numpy(b).cuda())
    ou, ov, x, y, oval = O.pass1(a, b, 32, 16, validate=True)
    w, o = 32, 16
    for p in range(3):
        pu, pv, pi = plan.pass_fields(p, 1) if p < 2 else (u, v, inv)
        if p > 0:
            w, o = w // 2, o // 2
            it = O.ITER["CWS"](a.shape, w, o)
            ou, ov, x, y, oval, du, dv, u0, v0, u2, v2 = it(a, b, x, y, ou.copy(), ov.copy(), oval.copy(), debug=True)
        err = np.maximum(np.abs(pu[0].cpu().numpy() - ou), np.abs(pv[0].cpu().numpy() - ov))
        flips = pi[0].cpu().numpy().astype(bool) != oval
        print(prec, "pass", p, "ws", w, "max err", err.max(), "n>1e-3", int((err > 1e-3).sum()), "flips", int(flips.sum()),
              "invalid", int(oval.sum()), np.argwhere(flips | (err > 1e-3))[:10].tolist())
        if p == 2:
            idx = O.window_index((H, W), w, o)
            f = (lambda t, dt: t.reshape(-1)[:, None, None].astype(dt))
            aa = O.shift_cws(a, idx, -f(u2, np.float32), -f(v2, np.float32)); bb = O.shift_cws(b, idx, f(u2, np.float32), f(v2, np.float32))
            nr, nc = ou.shape
            for ul in (16., 4096., 65536.):
                E = fp32_noise_excuse(aa, bb, nr, nc, ulps=ul)
                print("   ulps", ul, "excusable", int(E.sum()), "covering diff cells:", int((E & (flips | (err > 1e-3))).sum()))
            bad = np.argwhere(flips | (err > 1e-3))
            c = O.xcorr_fft(aa, bb); c = c - c.min(axis=(-2, -1), keepdims=True) + 1e-7
            for (r, cc) in bad[:6]:
                k = r * nc + cc
                flat = np.sort(c[k].ravel())
                m = c[k].ravel().argmax(); m2 = O.second_peak(c[k].reshape(1, -1).copy(), np.array([m]), 3, w, w)[0]
                print("    cell", r, cc, "ours", pu[0, r, cc].item(), pv[0, r, cc].item(), pi[0, r, cc].item(), "oracle", ou[r, cc], ov[r, cc], oval[r, cc],
                      "ratio", c[k].ravel()[m] / c[k].ravel()[m2], "top2", flat[-1], flat[-2], "u0", u0[r, cc], v0[r, cc], "du", du[r, cc], dv[r, cc])
    plan.close()
