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


def explained_region(a, b, ws, ov, n_pass, mode):
    """Cells of the YIELDED field (after hole fill and flip) that may legitimately differ from the
    reference: everything downstream of a window whose discrete decisions are a coin toss.
      * pass 1: exact arg-max ties and constant-input windows (black / saturated blocks);
      * pass p: the float32 noise band of the reference's own transform (fp32_noise_excuse, wide form
        for the float32 pass 1, including ill-conditioned sub-pixel fits: a peak neighbour at the map
        minimum), constant-input windows, and the cells whose spline predictor weight to an excusable
        coarse cell is >= 1e-4;
      * post-validation: a connected patch of (invalid or excusable) cells that contains an excusable
        cell may re-triangulate as a whole, so the patch and its ring are explained; so is a border
        edge that holds such a cell (1-D interpolation along the edge).
    Computed with the CPU oracle (pinned to the reference on these very frames by
    tests/test_oracle_golden.py::test_generator)."""
    from scipy import ndimage
    from oracle import piv_oracle as O
    from test_gpu_parity import constant_windows, fp32_noise_excuse, mask_ties, near_tie_windows, pass1_constant
    from torchpiv_amd import engine
    H, W = a.shape
    u, v, x, y, val = O.pass1(a, b, ws, ov, validate=True)
    E = near_tie_windows(a, b, ws, ov) | pass1_constant(a, b, ws, ov)
    # rounding noise in place of an exactly-zero fit (identical windows): its sign is a coin toss for the next pass
    E |= (np.maximum(np.abs(u), np.abs(v)) < 1e-9) & ((u != 0) | (v != 0))
    n_const = int(pass1_constant(a, b, ws, ov).sum())
    w, o = ws, ov
    for p in range(1, n_pass):
        xc, yc = x[0, :].copy(), y[:, 0].copy()
        ties = mask_ties(a.shape, (w, o), (w // 2, o // 2), val)       # predictor mask ON the 0.5 threshold
        w, o = w // 2, o // 2
        it = O.ITER[mode](a.shape, w, o)
        u, v, x, y, val, _, _, _, _, u2, v2 = it(a, b, x, y, u.copy(), v.copy(), val.copy(), debug=True)
        Ay, Ax = engine.spline_matrix(yc, y[:, 0]), engine.spline_matrix(xc, x[0, :])
        D = (np.abs(Ay) @ E.astype(np.float64) @ np.abs(Ax).T) >= 1e-4
        idx = O.window_index((H, W), w, o)
        f = (lambda t, dt: t.reshape(-1)[:, None, None].astype(dt))
        if mode == "CWS":
            aa = O.shift_cws(a, idx, -f(u2, np.float32), -f(v2, np.float32))
            bb = O.shift_cws(b, idx, f(u2, np.float32), f(v2, np.float32))
        else:
            aa = O.shift_dws(a, idx, -f(u2, np.int64), -f(v2, np.int64))
            bb = O.shift_dws(b, idx, f(u2, np.int64), f(v2, np.int64))
        nr, nc = u.shape
        E = D | fp32_noise_excuse(aa, bb, nr, nc, ulps=16.0, fit_tol=0.5e-3) | constant_windows(aa, bb, nr, nc)
        E |= ties
        n_const += int(constant_windows(aa, bb, nr, nc).sum())
    patch, n = ndimage.label(val | E)
    hit = np.unique(patch[E])
    region = np.isin(patch, hit[hit > 0]) | E
    region = ndimage.binary_dilation(region, structure=ndimage.generate_binary_structure(2, 1))
    for sl in ((0, slice(None)), (-1, slice(None)), (slice(None), 0), (slice(None), -1)):
        if region[sl].any():
            region[sl] = True
    return np.flip(region, axis=0), n_const


REGION_CAP = 0.25       # a comparison whose explained region covers more of the field than this checks nothing


def strict_chain(a, b, ws, ov, n_pass, mode, precision, unit, got_u, got_v, name, reference_chain=True):
    """What the explained-region comparison cannot see (frames with black / saturated blocks explain most of
    their small grids) is covered by a chain without a region: (i) the generator's tuple equals the REFERENCE's
    post-processing (oracle restatement of B:884-898: NaN-out, border interpolation, Delaunay fill, flip, scale)
    applied to the plan's own last-pass fields, everywhere, to 1e-9 px; (ii) those fields, and every pass before
    them, pass the three gates of cascade_check against the oracle's chain on the same frames."""
    from oracle import piv_oracle as O
    from test_gpu_fullsize import _oracle_fields
    from test_gpu_parity import cascade_check
    from torchpiv_amd import engine
    H, W = a.shape
    plan = engine.Plan(H, W, ws, ov, n_pass=n_pass, mode=mode, max_batch=1, precision=precision)
    geo = [tuple(t[:2]) for t in plan.geometry]
    u, v, inv = plan.run(torch.from_numpy(a).cuda(), torch.from_numpy(b).cuda())
    u, v, val = u[0].cpu().numpy(), v[0].cpu().numpy(), inv[0].cpu().numpy().astype(bool)
    plan.close()
    u[val] = np.nan
    v[val] = np.nan
    u, v = O.fill_missing(O.interp_borders(u)), O.fill_missing(O.interp_borders(v))
    assert u is not None and v is not None, name
    wu, wv = np.flip(u, axis=0) * unit, -np.flip(v, axis=0) * unit
    assert np.allclose(got_u, wu, rtol=0, atol=1e-9 * unit, equal_nan=True), name
    assert np.allclose(got_v, wv, rtol=0, atol=1e-9 * unit, equal_nan=True), name
    g = _oracle_fields(a, b, geo, mode, name)
    cascade_check(engine, g, name, mode, precision, geo, check_drift=reference_chain, strict_reference_chain=reference_chain)


# Pair 3 of the fixture is frame_b == frame_a without noise: the exact first-pass fit is 0, and the reference returns 0.0
# in all but two cells, where it returns its transform's rounding noise (1.8e-15).  The sign / zero-ness of such a value
# decides how the reference shifts the next pass's windows (PIVbackend.py:170, 193: an exactly integral coordinate takes
# another branch than one a rounding error away, and the flat-index wrap turns the sign of a 1e-17 px shift at column 0
# into another pixel): ANY non-zero spline weight of a noise cell turns an exactly-zero predictor into a non-zero one, and
# the reference's own multipass output for this pair reaches 3 px of "displacement" between identical frames.  It is a
# function of the reference's rounding noise, reproducible by nothing else, at any precision.  (finalize_kernel returns
# the exact 0 for identical windows, so this build's output for the pair is the clean one.)  What IS checked for it:
# pairs dropped / yielded, coordinates, shapes, the post-processing chain and the isolation gate of every pass
# (strict_chain with reference_chain=False); single-pass runs compare normally.
DEGENERATE = {("r2", 3), ("r3", 3), ("r4", 3)}


@pytest.mark.parametrize("precision", ["fast", "f64", "reference", "exact"])
@pytest.mark.parametrize("run", ["r1", "r2", "r3", "r4"])
def test_offline_piv_generator(folder, golden, run, precision):
    """The generator against the reference's own OfflinePIV runs: same pairs dropped, same coordinates,
    and every cell within 1e-3 px unless it is downstream of a coin-toss window (explained_region) --
    no fraction threshold."""
    import torchpiv_amd as T
    g = golden("g5_generator")
    ws, ov, mp_, mode, dt = (int(t) for t in g[run + "_kw"])
    scale = float(g[run + "_scale"][0])
    piv = T.OfflinePIV(folder, "cuda:0", "bmp", ws, ov, multipass=mp_, multipass_mode=("DWS", "CWS")[mode],
                       dt=dt, scale=scale, precision=precision)
    n_all, n_yield = (int(t) for t in g[run + "_count"])
    assert len(piv) == n_all
    res = list(piv())
    assert len(res) == n_yield, "same pairs must be dropped as in the reference"
    yielded = [i for i, *_ in piv.batched(batch_size=4)]
    for j, (x, y, u, v) in enumerate(res):
        assert u.dtype == np.float64 and x.dtype == np.float64
        assert np.array_equal(x, g[f"{run}_{j}_x"]) and np.array_equal(y, g[f"{run}_{j}_y"])
        assert u.shape == g[f"{run}_{j}_u"].shape
        unit = 1000 * scale / dt
        if (run, yielded[j]) in DEGENERATE:
            print(f"{run} {precision} pair {yielded[j]}: frame b == frame a -- NOT compared with the reference's chain (its output for this "
                  "pair is a function of +-1e-17 of transform noise, README 'Tolerance'); isolation gate of every pass only")
            strict_chain(g["frames_a"][yielded[j]], g["frames_b"][yielded[j]], ws, ov, mp_, ("DWS", "CWS")[mode],
                         precision, unit, u, v, f"{run}p{yielded[j]}", reference_chain=False)
            continue
        bad = ~np.isclose(u / unit, g[f"{run}_{j}_u"] / unit, rtol=0, atol=1e-3, equal_nan=True)
        bad |= ~np.isclose(v / unit, g[f"{run}_{j}_v"] / unit, rtol=0, atol=1e-3, equal_nan=True)
        i = yielded[j]
        region, n_const = explained_region(g["frames_a"][i], g["frames_b"][i], ws, ov, mp_, ("DWS", "CWS")[mode])
        print(f"{run} {precision} pair {i}: {int(bad.sum())} of {bad.size} cells beyond 1e-3 px, "
              f"{int((bad & ~region).sum())} unexplained; explained region covers {region.mean():.2f} of the field "
              f"({n_const} constant-input windows)")
        assert not (bad & ~region).any(), (run, j, np.argwhere(bad & ~region)[:6].tolist())
        # the region must leave something to compare -- unless the frames hold black / saturated blocks, whose
        # coin-toss windows reach most of these small grids through predictor and hole fill
        assert region.mean() <= REGION_CAP or n_const > 0, (run, i, float(region.mean()))
        strict_chain(g["frames_a"][i], g["frames_b"][i], ws, ov, mp_, ("DWS", "CWS")[mode], precision, unit, u, v,
                     f"{run}p{i}")
    # the batched extension gives the same tuples, tagged with the pair index
    res_b = list(piv.batched(batch_size=3))
    assert len(res_b) == len(res)
    for (i, x, y, u, v), (x2, y2, u2, v2) in zip(res_b, res):
        assert np.array_equal(u, u2, equal_nan=True) and np.array_equal(v, v2, equal_nan=True)
    print("  post-validation:", piv.stats)


def test_generator_config0_geometry(tmp_path, golden):
    """BASELINE.json configs[0]'s geometry (64/32, ONE pass, DWS) -- the bundled test_images are absent
    from the reference checkout, so the reference's run on seeded synthetic BMPs is the golden (g8 r5)."""
    from PIL import Image
    import torchpiv_amd as T
    g = golden("g8_round2")
    for i, (a, b) in enumerate(zip(g["r5_frames_a"], g["r5_frames_b"])):
        Image.fromarray(a, "L").save(tmp_path / f"image{8 + i}_a.bmp")
        Image.fromarray(b, "L").save(tmp_path / f"image{8 + i}_b.bmp")
    for precision in ("fast", "f64", "reference"):
        piv = T.OfflinePIV(str(tmp_path), "cuda:0", "bmp", 64, 32, multipass=1, multipass_mode="DWS", precision=precision)
        res = list(piv())
        assert len(piv) == int(g["r5_count"][0]) and len(res) == int(g["r5_count"][1]) > 0
        yielded = [i for i, *_ in piv.batched(batch_size=4)]
        for j, (x, y, u, v) in enumerate(res):
            assert np.array_equal(x, g[f"r5_{j}_x"]) and np.array_equal(y, g[f"r5_{j}_y"])
            bad = ~np.isclose(u / 1000, g[f"r5_{j}_u"] / 1000, rtol=0, atol=1e-3, equal_nan=True)
            bad |= ~np.isclose(v / 1000, g[f"r5_{j}_v"] / 1000, rtol=0, atol=1e-3, equal_nan=True)
            i = yielded[j]
            region, n_const = explained_region(g["r5_frames_a"][i], g["r5_frames_b"][i], 64, 32, 1, "DWS")
            print(f"r5 {precision} pair {i}: {int(bad.sum())} of {bad.size} cells beyond 1e-3 px, "
                  f"{int((bad & ~region).sum())} unexplained; explained region {region.mean():.2f} ({n_const} constant-input windows)")
            assert not (bad & ~region).any()
            assert region.mean() <= REGION_CAP or n_const > 0, (i, float(region.mean()))
            strict_chain(g["r5_frames_a"][i], g["r5_frames_b"][i], 64, 32, 1, "DWS", precision, 1000.0, u, v, f"r5p{i}")


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
        # exactly what the engine-level calls give (their parity with the reference is the strict per-pass test
        # test_iteration_golden_per_pass): same operators, same kernels, nothing in between
        from torchpiv_amd import engine as E
        H_, W_ = a.shape
        xc, yc = E.coordinates_1d(H_, W_, ws, ov)
        xf, yf = E.coordinates_1d(H_, W_, ws // 2, ov // 2)
        dv = lambda t: torch.from_numpy(np.ascontiguousarray(t)).cuda()
        pu0, pv0, pu2, pv2 = E.predict(mode, dv(E.spline_matrix(yc, yf)), dv(E.spline_matrix(xc, xf)),
                                       dv(g[f"{name}_{mode}_p0_u"])[None], dv(g[f"{name}_{mode}_p0_v"])[None],
                                       dv(g[f"{name}_{mode}_p0_val"].astype(np.uint8))[None])
        eu, ev, einv = E.iterate(mode, a, b, ws // 2, ov // 2, pu0, pv0, pu2, pv2)
        assert np.array_equal(u2, eu[0].cpu().numpy()) and np.array_equal(v2, ev[0].cpu().numpy())
        assert np.array_equal(val2, einv[0].cpu().numpy().astype(bool))
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
    # exactly the generator's fields (whose parity with the reference test_offline_piv_generator checks), in order
    piv2 = T.OfflinePIV(folder, "cuda:0", "bmp", 32, 16, multipass=3, multipass_mode="CWS")
    res = list(piv2())
    assert len(res) == 4
    for k, (gx, gy, gu, gv) in enumerate(res):
        assert np.array_equal(x, gx) and np.array_equal(y, gy)
        assert np.array_equal(uv[k, 0], gu, equal_nan=True) and np.array_equal(uv[k, 1], gv, equal_nan=True)
    piv.close()
    piv2.close()


def test_generator_reads_ahead_like_the_one_pair_loop(tmp_path):
    """__call__ runs `call_batch` pairs per launch; what it yields must be what the reference-literal loop
    (call_batch = 1) yields: same pairs, same order, same fields -- also when a file is undecodable (pair
    skipped, B:138-139), when a pair has another frame shape (its own plan) and when a batch holds nothing
    stageable."""
    from PIL import Image
    import torchpiv_amd as T
    from torchpiv_amd import synth
    shapes = [(512, 640)] * 3 + [(448, 576)] + [(512, 640)] * 3 + [(448, 576), (448, 576)]
    for i, (h, w) in enumerate(shapes):
        a, b = synth.make_pair(h, w, 70 + i, kind=("wavy", "vortex", "shear")[i % 3], noise=2.0)
        a, b = a.numpy().copy(), b.numpy().copy()
        a[140:170, 150:220] = 0         # dead windows: the pair has invalid vectors and is not dropped as "clean"
        b[140:170, 150:220] = 0
        Image.fromarray(a, "L").save(tmp_path / f"im{i:02d}_a.bmp")
        Image.fromarray(b, "L").save(tmp_path / f"im{i:02d}_b.bmp")
    (tmp_path / "im01_b.bmp").write_bytes(b"BMnot an image at all")          # undecodable: pair 1 is skipped
    kw = dict(wind_size=32, overlap=16, multipass=1, multipass_mode="CWS", scale=0.5, dt=2)

    def run(call_batch):
        piv = T.OfflinePIV(str(tmp_path), "cuda:0", "bmp", **kw)
        piv.call_batch = call_batch
        out = [tuple(np.array(t) for t in r) for r in piv()]
        piv.close()
        return out

    ref = run(1)
    assert len(ref) >= 4 and len({r[2].shape for r in ref}) == 2          # (dropped pairs aside) both shapes yield
    for cb in (2, 4, 16):
        got = run(cb)
        assert len(got) == len(ref)
        for g_, r_ in zip(got, ref):
            assert all(np.array_equal(x, y, equal_nan=True) for x, y in zip(g_, r_)), cb


@pytest.mark.gpu
def test_batched_over_png_files_and_abandoned_generators(tmp_path):
    """Formats the device cannot unpack never enter the native read-ahead ring (ReadAhead(read=False)): every file is
    decoded on the host and the batch still runs as one launch -- same tuples as the one-pair loop.  A generator that
    is abandoned half way must stop its reader threads (close()), and the next run over the same object starts clean."""
    from PIL import Image
    import torchpiv_amd as T
    from torchpiv_amd import synth
    for i in range(7):
        a, b = synth.make_pair(256, 320, 90 + i, kind=("wavy", "vortex", "shear")[i % 3], noise=2.0)
        a, b = a.numpy().copy(), b.numpy().copy()
        a[100:130, 90:160] = 0
        b[100:130, 90:160] = 0
        for fmt in ("png", "bmp"):
            Image.fromarray(a, "L").save(tmp_path / f"im{i:02d}_a.{fmt}")
            Image.fromarray(b, "L").save(tmp_path / f"im{i:02d}_b.{fmt}")
    kw = dict(wind_size=32, overlap=16, multipass=2, multipass_mode="DWS")
    out = {}
    for fmt in ("png", "bmp"):
        piv = T.OfflinePIV(str(tmp_path), "cuda:0", fmt, **kw)
        piv.call_batch = 1
        ref = [tuple(np.array(t) for t in r) for r in piv()]
        piv.call_batch = 3                                  # the reference's API: fresh, writable x, y per pair (B:899-900)
        g2 = piv()
        t0 = next(g2)
        assert t0[0].flags.writeable and t0[1].flags.writeable
        t0[0][:] = 0.0
        t1 = next(g2)
        assert t1[0].any() and t1[0] is not t0[0]
        g2.close()
        gen = piv.batched(3)
        first = next(gen)                                   # abandon the generator with batches still to come
        gen.close()
        rd = piv._reader
        assert rd._h is None                                # reader threads stopped and joined
        # a second abandoned run over the pairs in REVERSE order leaves uploads / unpack kernels of other files queued:
        # the restart below must not see their bytes (the staging and device buffers are reused)
        piv.pipeline_depth = 4
        gen = piv.batched(2, indices=list(range(len(piv) - 1, -1, -1)))
        next(gen)
        gen.close()
        piv.pipeline_depth = 2
        got = [tuple(np.array(t) for t in r[1:]) for r in piv.batched(3)]
        assert len(got) == len(ref) >= 5 and np.array_equal(first[3], ref[0][2], equal_nan=True)
        for g_, r_ in zip(got, ref):
            assert all(np.array_equal(x, y, equal_nan=True) for x, y in zip(g_, r_)), fmt
        out[fmt] = got
        piv.close()
    # and the two formats hold the same pixels, so they give the same fields
    for g_, r_ in zip(out["png"], out["bmp"]):
        assert all(np.array_equal(x, y, equal_nan=True) for x, y in zip(g_, r_))


@pytest.mark.gpu
def test_device_out_rows_equal_the_numpy_tuples(folder, golden):
    """device_out (what dist.run_sharded gathers from): batched() hands out rows of the per-batch DEVICE stacks of the
    finished fields, hole fills of the host stage scattered in -- the same bits as the numpy tuples of the default mode,
    for pairs finished on the device and for pairs that needed the host triangulation alike."""
    import torchpiv_amd as T
    g = golden("g5_generator")
    ws, ov, mp_, mode, dt = (int(t) for t in g["r1_kw"])
    piv = T.OfflinePIV(folder, "cuda:0", "bmp", ws, ov, multipass=mp_, multipass_mode=("DWS", "CWS")[mode], dt=dt,
                       scale=float(g["r1_scale"][0]))
    ref = {i: (u, v) for i, x, y, u, v in piv.batched(3)}
    piv.device_out = True
    got = {i: (u, v) for i, x, y, u, v in piv.batched(3)}
    assert ref and sorted(ref) == sorted(got) and piv.stats["host_fallback"] > 0
    for i in ref:
        assert isinstance(got[i][0], torch.Tensor) and got[i][0].is_cuda and got[i][0].dtype == torch.float64
        assert np.array_equal(got[i][0].cpu().numpy(), ref[i][0], equal_nan=True)
        assert np.array_equal(got[i][1].cpu().numpy(), ref[i][1], equal_nan=True)
    piv.close()


@pytest.mark.gpu
@pytest.mark.timeout(600)
def test_sharded_generator_rehearsal_two_ranks_one_gpu():
    """`bench.py --gpus 2 --e2e`: the generator path sharded over two ranks (gloo: both ranks time-slice the one GPU of
    the test box -- a rehearsal of the launch, of the per-rank host budget and of the single end-of-run gather from the
    device-resident fields, not a measurement), one JSON line with the `distributed` and `host` blocks."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, TPIV_DIST_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    env.pop("WORLD_SIZE", None)
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--e2e", "--e2e-pairs", "16",
                        "--size", "1024"], env=env, capture_output=True, text=True, timeout=560)
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads([l for l in r.stdout.strip().splitlines() if l.startswith("{")][-1])
    assert line["n_gpus"] == 2 and line["distributed"]["world_size"] == 2 and line["distributed"]["backend"] == "gloo"
    assert line["config"]["pairs_total"] == 2 * 16 * 4
    assert line["config"]["gathered_on_rank0"] == line["config"]["yielded_total"] > 0
    assert line["host"]["host_cpu_s_per_pair"] > 0 and line["host"]["budget_per_rank"]["ranks_on_node"] == 2
    assert sum(rk["post_validation"]["pairs"] for rk in line["distributed"]["ranks"]) == 2 * 16 * 4
