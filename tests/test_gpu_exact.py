"""precision="exact" (csrc/xcorr_exact.hip; every even first-pass window size from 8 to 128): pass 1 from exact integer
correlation sums, located by a float32 FFT pass, with the float64 transform for the windows that pass cannot decide.

Gates: (1) against the oracle's float64 pass 1 and the numpy statement of the scheme (tests/test_exact_scheme.py),
(2) against the float64 kernel of this library on full-size frames: fields within 1e-11 px, identical validity masks,
(3) the edge cases the scheme has branches for: dead windows, byte-identical frames, flat windows (undecidable by
construction -> float64 path), saturated frames, validation windows of other sizes, (4) the share of windows that take
the float64 path stays small on particle images, and the whole chain behind it gives the same final fields.
"""
import numpy as np
import pytest
import torch

from oracle import piv_oracle as O
from test_exact_scheme import exact_window, synthetic_pair

pytestmark = pytest.mark.gpu

TOL_F64 = 1e-11          # px, exact sums against a float64 transform (rounding of the transform through the fit)


@pytest.fixture(scope="module")
def eng():
    assert torch.cuda.is_available(), "GPU tests need a ROCm device"
    from torchpiv_amd import engine
    return engine


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def fields(eng, A, B, precision, ws=64, **kw):
    u, v, inv = eng.pass1(dev(A), dev(B), ws, kw.pop("ov", ws // 2), precision=precision, **kw)
    torch.cuda.synchronize()
    return u.cpu().numpy(), v.cpu().numpy(), inv.cpu().numpy()


@pytest.mark.parametrize("ws", [8, 16, 32, 64, 128, 12, 28, 42, 48, 22, 96])
@pytest.mark.parametrize("seed,shift", [(1, (2.3, -1.6)), (2, (0.0, 0.0)), (3, (-7.4, 11.2))])
def test_exact_pass1_against_oracle_and_model(eng, seed, shift, ws):
    A, B = synthetic_pair(256, 256, seed, shift)
    if shift[1] > ws / 4:
        shift = (shift[0] * ws / 64, shift[1] * ws / 64)      # (keep the displacement inside the small windows)
        A, B = synthetic_pair(256, 256, seed, shift)
    u0, v0, _, _, m0 = O.pass1(A, B, ws, ws // 2, validate=True)
    u, v, inv = fields(eng, A[None], B[None], "exact", ws)
    # sparse small windows have peak neighbours at the map minimum, where the log fit amplifies the float64 transform's own
    # rounding (the value there is 1e-7 + 1e-16 noise): the strict float64 gates set those aside the same way
    tol = TOL_F64 if ws >= 32 else 1e-9
    from test_gpu_parity import near_tie_windows, pass1_constant
    excused = pass1_constant(A, B, ws, ws // 2) | near_tie_windows(A, B, ws, ws // 2, rel=64 * 2.0 ** -52)
    bad = ((np.abs(u[0] - u0) > tol) | (np.abs(v[0] - v0) > tol) | (inv[0].astype(bool) != m0)) & ~excused
    assert not bad.any() and excused.mean() < 0.02, (ws, int(bad.sum()), int(excused.sum()), np.argwhere(bad)[:5].tolist())
    # the numpy statement of the scheme, window by window (its own float32 map may send other windows to the fallback)
    if ws in (32, 64, 128):
        aw, bw = O.windows(A, ws, ws // 2), O.windows(B, ws, ws // 2)
        for i, (a, b) in list(enumerate(zip(aw, bw)))[:: max(1, len(aw) // 50)]:
            r = exact_window(a, b)
            if r is not None:
                assert abs(r[0] - u[0].reshape(-1)[i]) < 1e-13 and abs(r[1] - v[0].reshape(-1)[i]) < 1e-13, i


@pytest.mark.parametrize("ws,ov", [(64, 32), (64, 48), (64, 10), (64, 0), (32, 16), (32, 7), (128, 64), (128, 100),
                                   (16, 8), (16, 3), (8, 4), (8, 0), (24, 12), (42, 21), (56, 5), (100, 50), (126, 63), (10, 5)])
def test_exact_equals_float64_kernel_on_full_frames(eng, ws, ov):
    from torchpiv_amd import synth
    A, B = synth.make_batch(4, 1024, 1024, device="cuda", noise=3.0, first_index=ov + 1)
    ue, ve, ie = eng.pass1(A, B, ws, ov, precision="exact")
    uf, vf, i_f = eng.pass1(A, B, ws, ov, precision="f64")
    assert float((ue - uf).abs().max()) < TOL_F64 and float((ve - vf).abs().max()) < TOL_F64
    assert torch.equal(ie, i_f)


@pytest.mark.parametrize("val_win,val_ratio", [(1, 1.2), (2, 1.05), (4, 1.5), (5, 1.2), (3, 3.0)])
def test_exact_validation_parameters(eng, val_win, val_ratio):
    from torchpiv_amd import synth
    A, B = synth.make_batch(2, 512, 512, device="cuda", noise=6.0, first_index=11)
    ue, ve, ie = eng.pass1(A, B, 64, 32, val_ratio=val_ratio, val_win=val_win, precision="exact")
    uf, vf, i_f = eng.pass1(A, B, 64, 32, val_ratio=val_ratio, val_win=val_win, precision="f64")
    assert float((ue - uf).abs().max()) < TOL_F64 and float((ve - vf).abs().max()) < TOL_F64
    assert torch.equal(ie, i_f)


@pytest.mark.parametrize("ws", [32, 64, 128])
def test_exact_edge_windows(eng, ws):
    rng = np.random.default_rng(3)
    H = W = 4 * ws
    A = rng.integers(0, 256, (5, H, W), dtype=np.uint8)
    B = rng.integers(0, 256, (5, H, W), dtype=np.uint8)
    A[0, ws:3 * ws, ws:3 * ws] = 0            # dead windows in frame a (zero mean: NaN map in the reference)
    B[1, :3 * ws // 2, :] = 0                 # ... in frame b
    B[2] = A[2]                               # byte-identical frames: the exact fit is 0
    A[3, ws // 2:5 * ws // 2, ws // 2:5 * ws // 2] = 77      # flat windows: constant map, nothing to decide -> float64 path
    B[3, ws // 2:5 * ws // 2, ws // 2:5 * ws // 2] = 91
    A[4] = 255                                # saturated frames
    B[4] = 255
    ue, ve, ie = fields(eng, A, B, "exact", ws)
    uf, vf, i_f = fields(eng, A, B, "f64", ws)
    assert np.abs(ue - uf).max() < TOL_F64 and np.abs(ve - vf).max() < TOL_F64
    assert np.array_equal(ie, i_f)
    assert np.all(ue[2] == 0.0) and np.all(ve[2] == 0.0)
    from test_gpu_parity import near_tie_windows, pass1_constant
    for k in (0, 1, 3):                       # and against the oracle itself, every window: the excuse set of the float64 gates --
        # constant windows (0/0 or all-rounding maps) and maps whose two largest float64 cells agree to 64 ulp
        u0, v0, _, _, m0 = O.pass1(A[k], B[k], ws, ws // 2, validate=True)
        excused = pass1_constant(A[k], B[k], ws, ws // 2) | near_tie_windows(A[k], B[k], ws, ws // 2, rel=64 * 2.0 ** -52)
        bad = ((np.abs(ue[k] - u0) > 1e-9) | (np.abs(ve[k] - v0) > 1e-9) | (ie[k].astype(bool) != m0)) & ~excused
        print(f"  edge windows ws {ws} frame {k}: {int(excused.sum())} of {excused.size} excused (constant / float64 tie), {int(bad.sum())} differing")
        assert not bad.any(), (k, np.argwhere(bad)[:5].tolist())


def test_exact_fallback_share_and_whole_chain(eng):
    """Noise-only and real-looking particle images: few windows need the float64 transform; the multipass chain behind an
    exact first pass ends in the same fields as behind the float64 first pass."""
    from torchpiv_amd import synth
    A, B = synth.make_batch(8, 1024, 1024, device="cuda", noise=2.0, first_index=5)
    pe = eng.Plan(1024, 1024, 64, 32, n_pass=2, mode="CWS", max_batch=8, precision="exact")
    pf = eng.Plan(1024, 1024, 64, 32, n_pass=2, mode="CWS", max_batch=8, precision="f64")
    ue, ve, ie = pe.run(A, B)
    n_fb = pe.exact_fallbacks()
    uf, vf, i_f = pf.run(A, B)
    n_win = 8 * 31 * 31
    assert 0 <= n_fb <= n_win // 20, (n_fb, n_win)
    # (a first-pass difference of 1e-14 px moves the second pass's bilinear samples by as much; a discrete decision of that
    #  pass -- arg-max, ratio threshold -- flips for a window only when it sits on the threshold to that precision)
    far = ((ue - uf).abs() > 1e-6) | ((ve - vf).abs() > 1e-6) | (ie != i_f)
    assert float(far.float().mean()) < 1e-3
    with pytest.raises(ValueError):
        pf.exact_fallbacks()


@pytest.mark.parametrize("ws,size,batch", [(32, 1024, 4), (128, 1024, 8)])
def test_exact_other_sizes_fallback_share(eng, ws, size, batch):
    from torchpiv_amd import synth
    A, B = synth.make_batch(batch, size, size, device="cuda", noise=2.0, first_index=17)
    pe = eng.Plan(size, size, ws, ws // 2, n_pass=1, max_batch=batch, precision="exact")
    assert "cand" in pe.kernel_name(0)
    ue, ve, ie = pe.run(A, B)
    n_fb, n_win = pe.exact_fallbacks(), batch * pe.geometry[0][2] * pe.geometry[0][3]
    uf, vf, i_f = eng.pass1(A, B, ws, ws // 2, precision="f64")
    print(f"ws {ws}: {n_fb} of {n_win} windows through the float64 transform")
    assert n_fb <= n_win // 20
    assert float((ue - uf).abs().max()) < TOL_F64 and float((ve - vf).abs().max()) < TOL_F64 and torch.equal(ie, i_f)



def test_exact_on_windows_built_to_break_the_locating_pass(eng):
    """tools/research/exact_band.py's families -- nearly orthogonal patterns (map range 1e-5 of the transform's scale), one
    bright pixel on a pedestal, two grey levels, saturated frames with a few dark pixels, sinusoids, ramps: whatever the
    float32 locating pass makes of them, the result must be the float64 kernel's (the contrast guard and the re-checks of
    the refinement send the hopeless ones to the float64 transform)."""
    import os
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "research"))
    import exact_band
    for name, (a, b) in exact_band.families(n=64, seed=1).items():
        A, B = dev(a), dev(b)
        plan = eng.Plan(64, 64, 64, 0, n_pass=1, max_batch=a.shape[0], precision="exact")
        ue, ve, ie = plan.run(A, B)
        n_fb = plan.exact_fallbacks()
        uf, vf, i_f = eng.pass1(A, B, 64, 0, precision="f64")
        d = max(float((ue - uf).abs().max()), float((ve - vf).abs().max()))
        print(f"  {name}: {n_fb} of {a.shape[0]} through the float64 transform, max |exact - f64| {d:.1e} px")
        assert d < TOL_F64 and torch.equal(ie, i_f), name
        if name.startswith("particles"):
            assert n_fb <= 2, (name, n_fb)
        if name == "particles on a true-zero background":
            assert n_fb <= a.shape[0] // 4, (name, n_fb)      # hundreds of minimum cells at S = 0 are decidable (S >= 0)
        if name == "checkerboard vs stripes":
            assert n_fb == a.shape[0], (name, n_fb)          # map range 8e-6 of the scale: below the contrast guard, all of them



def test_exact_random_geometries(eng):
    """Frame sizes that are not multiples of anything, any overlap, any batch, validation windows 1..5: precision "exact"
    against the float64 kernels on 40 seeded draws (row starts at every byte alignment, ragged last windows, one-window
    grids)."""
    rng = np.random.default_rng(20260)
    for case in range(60):
        ws = int(rng.choice([32, 64, 128, 8, 16, 12, 20, 30, 36, 40, 48, 26, 34, 72]))
        ov = int(rng.integers(0, ws))
        H = int(rng.integers(ws, 5 * ws + 7))
        W = int(rng.integers(ws, 5 * ws + 11))
        batch = int(rng.integers(1, 5))
        val_win = min(int(rng.integers(1, 6)), (ws - 1) // 2)      # (the library wants 2 val_win < ws)
        val_ratio = float(rng.choice([1.05, 1.2, 2.0]))
        # particle-like frames with a shift, plus a dead block and a saturated block now and then
        base = rng.integers(0, 40, (batch, H + 16, W + 16)).astype(np.float64)
        for _ in range(H * W // 60):
            y, x = int(rng.integers(2, H + 12)), int(rng.integers(2, W + 12))
            base[:, y - 1:y + 2, x - 1:x + 2] += rng.uniform(60, 200)
        sy, sx = int(rng.integers(0, 6)), int(rng.integers(0, 6))
        A = np.clip(base[:, 5:5 + H, 5:5 + W], 0, 255).astype(np.uint8)
        B = np.clip(base[:, 5 + sy:5 + sy + H, 5 + sx:5 + sx + W] + rng.integers(0, 6, (batch, H, W)), 0, 255).astype(np.uint8)
        if case % 5 == 0:
            A[0, : H // 3, : W // 2] = 0
        if case % 7 == 0:
            B[-1, H // 2:, W // 3:] = 255
        kw = dict(val_ratio=val_ratio, val_win=val_win)
        ue, ve, ie = eng.pass1(dev(A), dev(B), ws, ov, precision="exact", **kw)
        uf, vf, i_f = eng.pass1(dev(A), dev(B), ws, ov, precision="f64", **kw)
        d = max(float((ue - uf).abs().max()), float((ve - vf).abs().max()))
        assert d < TOL_F64 and torch.equal(ie, i_f), (case, ws, ov, H, W, batch, val_win, d, int((ie != i_f).sum()))



def test_exact_on_adversarial_windows(eng):
    """tests/golden/g12_adversarial.npz: the windows tools/research/exact_adversarial.py found by hill-climbing on the float32
    map's error relative to E+ (nine seed families, 64 x 64; worst ratio 8.0e-7 against the proven Gamma = 1.47e-5 the
    band is built from, profiles/r05/exact_adversarial.txt).  (1) the float32 map of the tile kernel stays inside Gamma E+ on
    every one of them, with an order of magnitude to spare; (2) precision "exact" gives the float64 kernels' fields."""
    import os
    from test_exact_scheme import band_coef, e_plus
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "g12_adversarial.npz"))
    worst = 0.0
    for i, name in enumerate(g["names"]):
        P = g[f"w{i}"]
        a, b = np.ascontiguousarray(P[:, 0]), np.ascontiguousarray(P[:, 1])
        A, B = dev(a), dev(b)
        _, _, _, _, corr = eng.debug_pass(0, A, B, 64, 0, precision="fast")
        c32 = corr.cpu().numpy().reshape(-1, 64, 64).astype(np.float64)
        af, bf = a.astype(np.float64), b.astype(np.float64)
        an = af / af.mean(axis=(1, 2), keepdims=True) - 1
        bn = bf / bf.mean(axis=(1, 2), keepdims=True) - 1
        c64 = np.fft.fftshift(np.fft.irfft2(np.conj(np.fft.rfft2(an)) * np.fft.rfft2(bn), s=(64, 64)), axes=(1, 2))
        e = (c32 - (c64 - c64.min(axis=(1, 2), keepdims=True) + 1e-7)).reshape(len(a), -1)
        err = 0.5 * (e.max(axis=1) - e.min(axis=1))            # (up to the common offset, which changes no decision)
        ep = np.array([e_plus(x, y) for x, y in zip(a, b)])
        ratio = float((err / ep).max())
        worst = max(worst, ratio)
        gamma = band_coef(64) / (2 * (1 + 1 / 16))
        assert ratio < gamma / 8, (str(name), ratio, gamma)
        plan = eng.Plan(64, 64, 64, 0, n_pass=1, max_batch=len(a), precision="exact")
        ue, ve, ie = plan.run(A, B)
        n_fb = plan.exact_fallbacks()
        plan.close()
        uf, vf, i_f = eng.pass1(A, B, 64, 0, precision="f64")
        d = max(float((ue - uf).abs().max()), float((ve - vf).abs().max()))
        print(f"  {str(name):32s} err / E+ {ratio:.2e}  float64 path {n_fb}/{len(a)}  max |exact - f64| {d:.1e} px")
        assert d < TOL_F64 and torch.equal(ie, i_f), str(name)
    print(f"  worst err / E+ {worst:.2e} against Gamma {gamma:.2e}")


@pytest.mark.parametrize("ws,scale,size,batch", [(16, 2.0, 1024, 4), (48, 1.5, 1024, 4), (24, 2.0, 1024, 4), (96, 2.0, 1024, 2)])
def test_exact_chains_from_other_first_pass_sizes(eng, ws, scale, size, batch):
    """multipass chains whose FIRST pass is not 32 / 64 / 128 pixels (16/8 -> 8/4, 48/24 -> 32/16 at
    multipass_scale 1.5, 24/12 -> 12/6, 96/48 -> 48/24) at the default precision against the float64 first pass: the same first-pass
    fields to 1e-9 px with identical flags, and final fields that differ in no more than a sliver of the cells."""
    from torchpiv_amd import synth
    A, B = synth.make_batch(batch, size, size, device="cuda", noise=2.0, first_index=200 + ws)
    pe = eng.Plan(size, size, ws, ws // 2, n_pass=2, mode="CWS", pass_scale=scale, max_batch=batch, precision="exact")
    pf = eng.Plan(size, size, ws, ws // 2, n_pass=2, mode="CWS", pass_scale=scale, max_batch=batch, precision="f64")
    assert "cand" in pe.kernel_name(0), pe.kernel_name(0)
    ue, ve, ie = (t.clone() for t in pe.run(A, B))
    n_fb, n_win = pe.exact_fallbacks(), batch * pe.geometry[0][2] * pe.geometry[0][3]
    u1e, v1e, i1e = pe.pass_fields(0, batch)
    uf, vf, i_f = (t.clone() for t in pf.run(A, B))
    u1f, v1f, i1f = pf.pass_fields(0, batch)
    d1 = max(float((u1e - u1f).abs().max()), float((v1e - v1f).abs().max()))
    print(f"  {ws}/{ws // 2} x{scale}: geometry {pe.geometry}, {n_fb} of {n_win} first-pass windows through the float64 transform, "
          f"first pass max |exact - f64| {d1:.1e} px")
    assert d1 < 1e-9 and torch.equal(i1e, i1f)
    assert n_fb <= n_win // 10
    far = ((ue - uf).abs() > 1e-6) | ((ve - vf).abs() > 1e-6) | (ie != i_f)
    assert float(far.float().mean()) < 2e-3
    pe.close()
    pf.close()
