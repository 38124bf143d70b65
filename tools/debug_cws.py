import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from oracle import piv_oracle as O
from torchpiv_amd import engine
g = np.load("tests/golden/g5_generator.npz")
a, b = g["frames_a"][3], g["frames_b"][3]
print("a==b:", np.array_equal(a, b))
H, W = a.shape
ws, ov = 64, 32
ou, ov_, x, y, oval = O.pass1(a, b, ws, ov, validate=True)
it = O.ITER["CWS"](a.shape, 32, 16)
ru, rv, _, _, rval, rdu, rdv, ru0, rv0, ru2, rv2 = it(a, b, x, y, ou.copy(), ov_.copy(), oval.copy(), debug=True)
dev = lambda t: torch.from_numpy(np.ascontiguousarray(t)).cuda()
u0, v0, u2, v2 = dev(ru0)[None], dev(rv0)[None], dev(ru2)[None], dev(rv2)[None]
gu, gv, ginv, gdu, gdv = engine.iterate("CWS", dev(a), dev(b), 32, 16, u0, v0, u2, v2, want_raw=True)
gdu, gdv = gdu[0].cpu().numpy(), gdv[0].cpu().numpy()
err = np.maximum(np.abs(gdu - rdu), np.abs(gdv - rdv))
print("given the ORACLE's predictor: max raw err", err.max(), "n>1e-3", (err > 1e-3).sum())
idx = np.argwhere(err > 1e-4)[:6]
for (r, c) in idx:
    print(r, c, "gpu du,dv", gdu[r, c], gdv[r, c], "ref", rdu[r, c], rdv[r, c], "u2,v2", ru2[r, c], rv2[r, c])
# staged windows + corr maps
_, _, _, win, corr = engine.debug_pass("CWS", dev(a), dev(b), 32, 16, u2, v2)
aa = O.shift_cws(a, it.idx, -ru2.astype(np.float32).reshape(-1)[:, None, None], -rv2.astype(np.float32).reshape(-1)[:, None, None])
bb = O.shift_cws(b, it.idx, ru2.astype(np.float32).reshape(-1)[:, None, None], rv2.astype(np.float32).reshape(-1)[:, None, None])
print("windows bit-exact:", np.array_equal(win[0, :, 0].cpu().numpy(), aa), np.array_equal(win[0, :, 1].cpu().numpy(), bb))
c = O.xcorr_fft(aa, bb); c = c - c.min(axis=(-2, -1), keepdims=True) + np.float32(1e-7)
c64 = O.xcorr_fft(aa.astype(np.float64), bb.astype(np.float64)); c64 = c64 - c64.min(axis=(-2, -1), keepdims=True) + 1e-7
gc = corr[0].cpu().numpy()
for (r, c_) in idx[:3]:
    n = r * it.n_cols + c_
    m = np.unravel_index(np.argmax(c64[n]), c64[n].shape)
    print("win", n, "peak", m, "peak val f64", c64[n][m], "f32ref", c[n][m], "gpu", gc[n][m])
    for d in ((0, 1), (0, -1), (1, 0), (-1, 0)):
        q = (m[0] + d[0], m[1] + d[1])
        print("    nb", d, "f64", c64[n][q], "ref32", c[n][q], "gpu", gc[n][q])
    du64, dv64, _ = O.corr_to_disp(c64[n:n+1] - 1e-7, 1, 1, True)
    print("    f64 fit:", du64, dv64)
