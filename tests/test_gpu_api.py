"""The drop-in boundary on the GPU: OfflinePIV generator, the function-level seam with the
reference's signatures, and the batched / sharded extensions -- against the goldens captured
from the reference's own OfflinePIV runs."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def folder(tmp_path_factory, golden):
    from PIL import Image
    g = golden("g5_generator")
    d = tmp_path_factory.mktemp("pairs")
    for i, (a, b) in enumerate(zip(g["frames_a"], g["frames_b"])):
        Image.fromarray(a, "L").save(os.path.join(d, f"image{8 + i}_a.bmp"))
        Image.fromarray(b, "L").save(os.path.join(d, f"image{8 + i}_b.bmp"))
    return str(d)


def _close(got, want, frac_ok=0.75, tol=1e-3):
    """Final fields after all passes and the hole fill (in px: callers divide by 1000*scale/dt).
    A flipped validity decision upstream (a window inside the float32 rounding band of the
    reference's own transform, see test_gpu_parity.fp32_noise_excuse) moves the predictor of the
    finer windows around it and re-triangulates the hole fill, so a handful of flips changes a
    whole patch of interpolated cells: numerically equivalent builds of the kernels have scored
    0.82 - 0.91 on the worst fixture pair.  The end-to-end criterion is therefore robust: the BULK
    must be exact (median error far below the tolerance) and at least `frac_ok` of the vectors
    within tolerance; strict per-pass parity is tests/test_gpu_parity.py."""
    ok = np.isclose(got, want, rtol=0, atol=tol, equal_nan=True)
    both = np.isfinite(got) & np.isfinite(want)
    med = np.median(np.abs(got[both] - want[both])) if both.any() else 0.0
    return ok.mean() >= frac_ok and med < 1e-5


# Pair 3 of the fixture is frame_b == frame_a without noise: its predictor is ~ +-1e-8 px and the
# reference's CWS "integral coordinate => nearest sample" quirk (PIVbackend.py:170,193) makes the
# result depend on the SIGN of that rounding noise -- not reproducible by any other arithmetic.
DEGENERATE = {("r2", 3), ("r4", 3)}


@pytest.mark.parametrize("run", ["r1", "r2", "r3", "r4"])
def test_offline_piv_generator(folder, golden, run):
    import torchpiv_amd as T
    g = golden("g5_generator")
    ws, ov, mp_, mode, dt = (int(t) for t in g[run + "_kw"])
    scale = float(g[run + "_scale"][0])
    piv = T.OfflinePIV(folder, "cuda:0", "bmp", ws, ov, multipass=mp_, multipass_mode=("DWS", "CWS")[mode],
                       dt=dt, scale=scale)
    n_all, n_yield = (int(t) for t in g[run + "_count"])
    assert len(piv) == n_all
    res = list(piv())
    assert len(res) == n_yield, "same pairs must be dropped as in the reference"
    yielded = [i for i, *_ in piv.batched(batch_size=4)]
    for j, (x, y, u, v) in enumerate(res):
        assert u.dtype == np.float64 and x.dtype == np.float64
        assert np.array_equal(x, g[f"{run}_{j}_x"]) and np.array_equal(y, g[f"{run}_{j}_y"])
        assert u.shape == g[f"{run}_{j}_u"].shape
        if (run, yielded[j]) in DEGENERATE:
            continue
        unit = 1000 * scale / dt
        fu = np.isclose(u / unit, g[f"{run}_{j}_u"] / unit, rtol=0, atol=1e-3, equal_nan=True).mean()
        fv = np.isclose(v / unit, g[f"{run}_{j}_v"] / unit, rtol=0, atol=1e-3, equal_nan=True).mean()
        print(f"{run} pair {yielded[j]}: within 1e-3 px: u {fu:.4f} v {fv:.4f}")
        assert _close(u / unit, g[f"{run}_{j}_u"] / unit), (run, j, fu)
        assert _close(v / unit, g[f"{run}_{j}_v"] / unit), (run, j, fv)
    # the batched extension gives the same tuples, tagged with the pair index
    res_b = list(piv.batched(batch_size=3))
    assert len(res_b) == len(res)
    for (i, x, y, u, v), (x2, y2, u2, v2) in zip(res_b, res):
        assert np.array_equal(u, u2, equal_nan=True) and np.array_equal(v, v2, equal_nan=True)


def test_function_seam_signatures(golden):
    """extended_search_area_piv / piv_iteration_X called the way the reference calls them."""
    import torchpiv_amd as T
    g = golden("g4_multipass")
    name = "special32x2"
    ws, ov, n_pass = (int(t) for t in g[name + "_cfg"])
    a = torch.from_numpy(g[name + "_a"]).cuda()
    b = torch.from_numpy(g[name + "_b"]).cuda()
    u, v, x, y, val = T.extended_search_area_piv(a, b, window_size=ws, overlap=ov, validate=True)
    assert isinstance(u, np.ndarray) and val.dtype == bool
    assert np.abs(u - g[f"{name}_CWS_p0_u"]).max() < 1e-3 and np.array_equal(val, g[f"{name}_CWS_p0_val"])
    u_, v_, x_, y_, none = T.extended_search_area_piv(a, b, window_size=ws, overlap=ov)
    assert none is None and np.array_equal(u_, u)
    for mode in ("DWS", "CWS"):
        it = T.IterModMap.functions[mode](a.shape, ws // 2, ov // 2, "cuda:0")
        u2, v2, x2, y2, val2 = it(a, b, x, y, g[f"{name}_{mode}_p0_u"].copy(), g[f"{name}_{mode}_p0_v"].copy(),
                                  g[f"{name}_{mode}_p0_val"].copy())
        same = val2 == g[f"{name}_{mode}_p1_val"]
        assert same.mean() > 0.99
        err = np.maximum(np.abs(u2 - g[f"{name}_{mode}_p1_u"]), np.abs(v2 - g[f"{name}_{mode}_p1_v"]))
        assert (err[same] < 1e-3).mean() > 0.99, mode
        assert np.array_equal(x2, it.x) and x2.shape == u2.shape
        # validation_mask=None: no peak-ratio test, val stays None (B:707-709)
        u3, v3, _, _, val3 = it(a, b, x, y, g[f"{name}_{mode}_p0_u"].copy(), g[f"{name}_{mode}_p0_v"].copy(), None)
        assert val3 is None and np.isfinite(u3).all()
    with pytest.raises(ValueError):
        T.extended_search_area_piv(a, b, window_size=32, overlap=32)


def test_run_sharded_single_process(folder, golden):
    import torchpiv_amd as T
    from torchpiv_amd import dist as pdist
    g = golden("g5_generator")
    piv = T.OfflinePIV(folder, "cuda:0", "bmp", 32, 16, multipass=3, multipass_mode="CWS")
    ids, (x, y), uv = pdist.run_sharded(piv, batch_size=2)
    assert ids.tolist() == [0, 1, 2, 3] and uv.shape[1] == 2
    assert _close(uv[0, 0] / 1000, g["r4_0_u"] / 1000) and _close(uv[1, 1] / 1000, g["r4_1_v"] / 1000)
