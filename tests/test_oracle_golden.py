"""Pin the CPU oracle (oracle/piv_oracle.py) to the reference's own outputs
(tests/golden/*.npz, made by tests/golden/make_golden.py from the reference)."""
import numpy as np
import pytest

from oracle import piv_oracle as O

TOL = 1e-12


def test_geometry(golden):
    g = golden("g1_geometry")
    for H, W, ws, ov in g["cases"]:
        key = f"{H}_{W}_{ws}_{ov}"
        assert np.array_equal(O.field_shape((H, W), ws, ov), g["fs_" + key])
        x, y = O.coordinates((H, W), ws, ov)
        assert np.array_equal(x[0, :], g["x_" + key])
        assert np.array_equal(y[:, 0], g["y_" + key])


def test_pass1(golden):
    g = golden("g3_pass1")
    for name in g["names"]:
        ws, ov = g[name + "_cfg"]
        u, v, x, y, mask = O.pass1(g[name + "_a"], g[name + "_b"], int(ws), int(ov), validate=True)
        assert np.abs(u - g[name + "_u"]).max() <= TOL, name
        assert np.abs(v - g[name + "_v"]).max() <= TOL, name
        assert np.array_equal(mask, g[name + "_mask"]), name


@pytest.mark.parametrize("mode", ["DWS", "CWS"])
def test_multipass(golden, mode):
    g = golden("g4_multipass")
    for name in g["names"]:
        ws, ov, n_pass = (int(t) for t in g[name + "_cfg"])
        a, b = g[name + "_a"], g[name + "_b"]
        u, v, x, y, val = O.pass1(a, b, ws, ov, validate=True)
        assert np.abs(u - g[f"{name}_{mode}_p0_u"]).max() <= TOL
        w, o = ws, ov
        for p in range(1, n_pass):
            w, o = int(w // 2.0), int(o // 2.0)
            it = O.ITER[mode](a.shape, w, o)
            u, v, x, y, val = it(a, b, x, y, u.copy(), v.copy(), val.copy())
            # float32 FFT runs through the same torch/MKL build as the reference here,
            # so the agreement is to rounding of the float64 epilogue
            assert np.abs(u - g[f"{name}_{mode}_p{p}_u"]).max() <= 1e-9, (name, p)
            assert np.abs(v - g[f"{name}_{mode}_p{p}_v"]).max() <= 1e-9, (name, p)
            assert np.array_equal(val, g[f"{name}_{mode}_p{p}_val"]), (name, p)


def test_generator(golden):
    g = golden("g5_generator")
    A, B = g["frames_a"], g["frames_b"]
    for r in ("r1", "r2", "r3", "r4"):
        ws, ov, mp, mode, dt = (int(t) for t in g[r + "_kw"])
        scale = float(g[r + "_scale"][0])
        res = list(O.offline_piv(zip(A, B), ws, ov, multipass=mp, mode=("DWS", "CWS")[mode],
                                 dt=dt, scale=scale))
        n_all, n_yield = g[r + "_count"]
        assert len(res) == n_yield, r
        for j, (x, y, u, v) in enumerate(res):
            assert np.array_equal(x, g[f"{r}_{j}_x"]) and np.array_equal(y, g[f"{r}_{j}_y"])
            assert np.allclose(u, g[f"{r}_{j}_u"], rtol=0, atol=1e-6, equal_nan=True), (r, j)
            assert np.allclose(v, g[f"{r}_{j}_v"], rtol=0, atol=1e-6, equal_nan=True), (r, j)


def test_kats_corr_to_disp(golden):
    g = golden("g6_kats")
    maps = g["c2d16_maps"]
    for dt, nm in ((np.float32, "f32"), (np.float64, "f64")):
        u, v, mask = O.corr_to_disp(maps.astype(dt), maps.shape[0], 1, validate=True)
        assert np.allclose(u, g[f"c2d16_{nm}_u"], rtol=0, atol=TOL), nm
        assert np.allclose(v, g[f"c2d16_{nm}_v"], rtol=0, atol=TOL), nm
        assert np.array_equal(mask, g[f"c2d16_{nm}_mask"]), nm
    u, v, mask = O.corr_to_disp(g["c2d8_maps"].astype(np.float32), g["c2d8_maps"].shape[0], 1, True)
    assert np.allclose(u, g["c2d8_u"], rtol=0, atol=TOL) and np.allclose(v, g["c2d8_v"], rtol=0, atol=TOL)
    assert np.array_equal(mask, g["c2d8_mask"])
    u, v, mask = O.corr_to_disp(g["c2dr_maps"].astype(np.float64), 6, 1, True)
    assert np.allclose(u, g["c2dr_u"], rtol=0, atol=TOL) and np.allclose(v, g["c2dr_v"], rtol=0, atol=TOL)
    assert np.array_equal(mask, g["c2dr_mask"])
    # known answers: a pure Gaussian at (col 6.8, row 9.3) is fitted exactly; edge peaks are
    # one-sided and give exactly +-0.5
    assert abs(g["c2d16_f64_u"][0, 0] - (6.8 - 8)) < 1e-8 and abs(g["c2d16_f64_v"][0, 0] - (9.3 - 8)) < 1e-8
    assert g["c2d16_f64_u"][1, 0] == -8.5 and g["c2d16_f64_v"][1, 0] == -8.5
    assert g["c2d16_f64_u"][2, 0] == 7.5 and g["c2d16_f64_v"][2, 0] == 7.5


def test_kats_shift_and_xcorr(golden):
    g = golden("g6_kats")
    frame = g["shift_frame"]
    ws, ov = g["shift_cfg"]
    idx = O.window_index(frame.shape, int(ws), int(ov))
    cws = O.shift_cws(frame, idx, g["shift_vx"][:, None, None], g["shift_vy"][:, None, None])
    assert cws.dtype == np.float32 and np.array_equal(cws, g["shift_cws"])
    dws = O.shift_dws(frame, idx, g["shift_ix"][:, None, None], g["shift_iy"][:, None, None])
    assert np.array_equal(dws, g["shift_dws"])
    c = O.xcorr_fft(g["xc_a"], g["xc_b"])
    assert c.dtype == np.float32 and np.allclose(c, g["xc_u8"], rtol=1e-6, atol=1e-2)
    c = O.xcorr_fft(g["xc_a"].astype(np.float64), g["xc_b"].astype(np.float64))
    assert np.allclose(c, g["xc_f64"], rtol=1e-13, atol=1e-7)


def test_kats_post_validation(golden):
    g = golden("g6_kats")
    gb = O.interp_borders(g["pv_in"].copy())
    assert np.allclose(gb, g["pv_borders"], rtol=0, atol=TOL, equal_nan=True)
    filled = O.fill_missing(gb.copy())
    assert np.allclose(filled, g["pv_filled"], rtol=0, atol=TOL, equal_nan=True)
    # quirk: a field with no invalid vector is dropped (returns None), B:303-304
    assert O.fill_missing(np.ones((5, 5))) is None


def test_generic_sizes(golden):
    """Non power-of-two windows and a 1.5 refinement scale (64 -> 42 -> 28), as the reference allows."""
    g = golden("g7_generic")
    for name in g["p1_names"]:
        ws, ov = (int(t) for t in g[name + "_cfg"])
        u, v, x, y, mask = O.pass1(g[name + "_a"], g[name + "_b"], ws, ov, validate=True)
        assert np.abs(u - g[name + "_u"]).max() <= TOL and np.abs(v - g[name + "_v"]).max() <= TOL, name
        assert np.array_equal(mask, g[name + "_mask"]), name
    for name in g["mp_names"]:
        a, b = g[name + "_a"], g[name + "_b"]
        geo = g[name + "_geo"]
        for mode in ("DWS", "CWS"):
            u, v, x, y, val = O.pass1(a, b, int(geo[0][0]), int(geo[0][1]), validate=True)
            for p in range(1, len(geo)):
                it = O.ITER[mode](a.shape, int(geo[p][0]), int(geo[p][1]))
                u, v, x, y, val = it(a, b, x, y, u.copy(), v.copy(), val.copy())
                assert np.abs(u - g[f"{name}_{mode}_p{p}_u"]).max() <= 1e-9, (name, mode, p)
                assert np.array_equal(val, g[f"{name}_{mode}_p{p}_val"]), (name, mode, p)


def test_round2_goldens(golden):
    """g8: 256/128 -> 128/64 multipass (the only route to shifted 128-pixel windows) and the
    generator at configs[0]'s geometry (64/32, one pass)."""
    g = golden("g8_round2")
    for name in ("big256x2", "odd66x2"):          # odd66x2: a shifted pass with an ODD window (33 x 32 map quirk)
        ws, ov, n_pass = (int(t) for t in g[name + "_cfg"])
        a, b = g[name + "_a"], g[name + "_b"]
        for mode in ("DWS", "CWS"):
            u, v, x, y, val = O.pass1(a, b, ws, ov, validate=True)
            assert np.abs(u - g[f"{name}_{mode}_p0_u"]).max() <= TOL
            it = O.ITER[mode](a.shape, ws // 2, ov // 2)
            u, v, x, y, val = it(a, b, x, y, u.copy(), v.copy(), val.copy())
            assert np.abs(u - g[f"{name}_{mode}_p1_u"]).max() <= 1e-9 and np.abs(v - g[f"{name}_{mode}_p1_v"]).max() <= 1e-9
            assert np.array_equal(val, g[f"{name}_{mode}_p1_val"])
    ws, ov, mp, mode, dt = (int(t) for t in g["r5_kw"])
    res = list(O.offline_piv(zip(g["r5_frames_a"], g["r5_frames_b"]), ws, ov, multipass=mp,
                             mode=("DWS", "CWS")[mode], dt=dt, scale=float(g["r5_scale"][0])))
    assert len(res) == int(g["r5_count"][1]) and len(res) > 0
    for j, (x, y, u, v) in enumerate(res):
        assert np.array_equal(x, g[f"r5_{j}_x"]) and np.array_equal(y, g[f"r5_{j}_y"])
        assert np.allclose(u, g[f"r5_{j}_u"], rtol=0, atol=1e-9, equal_nan=True)
        assert np.allclose(v, g[f"r5_{j}_v"], rtol=0, atol=1e-9, equal_nan=True)


@pytest.mark.parametrize("mode", ["DWS", "CWS"])
def test_other_scales_and_zero_overlap(golden, mode):
    """g11: multipass_scale 1.5 (64/32 -> 42/21 -> 28/14), 4.0 (64/32 -> 16/8) and zero overlap (32/0 -> 16/0)."""
    g = golden("g11_scales")
    for name in g["names"]:
        ws, ov, n_pass = (int(t) for t in g[name + "_cfg"])
        scale = float(g[name + "_scale"][0])
        a, b = g[name + "_a"], g[name + "_b"]
        u, v, x, y, val = O.pass1(a, b, ws, ov, validate=True)
        assert np.abs(u - g[f"{name}_{mode}_p0_u"]).max() <= TOL
        w, o = ws, ov
        for p in range(1, n_pass):
            w, o = int(w // scale), int(o // scale)
            assert [w, o] == [int(t) for t in g[name + "_geo"][p]]
            u, v, x, y, val = O.ITER[mode](a.shape, w, o)(a, b, x, y, u.copy(), v.copy(), val.copy())
            assert np.abs(u - g[f"{name}_{mode}_p{p}_u"]).max() <= 1e-9, (name, p)
            assert np.abs(v - g[f"{name}_{mode}_p{p}_v"]).max() <= 1e-9, (name, p)
            assert np.array_equal(val, g[f"{name}_{mode}_p{p}_val"]), (name, p)


def test_cws_fast_iteration(golden):
    """piv_iteration_CWS_Fast (B:599-675, unreachable from the reference's own OfflinePIV): the oracle's
    restatement against the reference's output."""
    g = golden("g10_cws_fast")
    for name in g["names"]:
        ws, ov = (int(t) for t in g[name + "_cfg"])
        a, b = g[name + "_a"], g[name + "_b"]
        x, y = O.coordinates(a.shape, ws, ov)
        it = O.IterCWSFast(a.shape, ws // 2, ov // 2)
        u, v, _, _, val = it(a, b, x, y, g[name + "_p0_u"].copy(), g[name + "_p0_v"].copy(), g[name + "_p0_val"].copy())
        assert np.abs(u - g[name + "_p1_u"]).max() <= 1e-9 and np.abs(v - g[name + "_p1_v"]).max() <= 1e-9, name
        assert np.array_equal(val, g[name + "_p1_val"]), name
