"""Round-2 parity additions (through the C ABI, on the GPU):
  * precision="reference": pass 1 in float64 like the reference (PIVbackend.py:513-514) -- goldens
    WITHOUT any excuse set;
  * shifted 128x128 passes (256/128 -> 128/64), which run the generic-size kernel;
  * the function-level seam takes caller-provided work buffers: calls on two streams may overlap.
"""
import numpy as np
import pytest
import torch

from oracle import piv_oracle as O
from test_gpu_parity import (TOL_PX, cascade_check, check_fields, constant_windows, dev, fp32_noise_excuse,
                             pass1_constant, staged_windows)

pytestmark = pytest.mark.gpu

# float64 pass 1 against the reference's float64 pass 1: both carry rounding noise of a few ulp of the DC
# pedestal (n^2 = 4096 for 64x64 windows) on every map value, which the log-ratio of the sub-pixel fit
# amplifies for windows whose peak neighbours are close to the map minimum.  Observed: <= 1e-10 px.
TOL_REF = 1e-9


@pytest.fixture(scope="module")
def eng():
    assert torch.cuda.is_available(), "GPU tests need a ROCm device"
    from torchpiv_amd import engine
    return engine


def exact_tie_windows(a, b, ws, ov):
    """Windows whose float64 map holds its maximum at two cells to within 64 ulp of the DC pedestal: an
    arg-max there is decided by the rounding of the transform itself (integer-valued noise windows do
    produce such ties), in the reference as in any other float64 implementation."""
    aa = O.windows(a, ws, ov).astype(np.float64)
    bb = O.windows(b, ws, ov).astype(np.float64)
    with np.errstate(all="ignore"):
        aa = aa / aa.mean(axis=(-2, -1), keepdims=True)
        bb = bb / bb.mean(axis=(-2, -1), keepdims=True)
    c = O.xcorr_fft(aa, bb)
    f = np.sort(c.reshape(c.shape[0], -1), axis=-1)
    with np.errstate(all="ignore"):
        tie = (f[:, -1] - f[:, -2]) <= 64 * 2.0 ** -52 * np.abs(c).max(axis=(-2, -1))
    nr, nc = O.field_shape(a.shape, ws, ov)
    return tie.reshape(nr, nc)


def check_reference_precision(eng, a, b, ws, ov, ru, rv, rmask, what, precision="reference"):
    u, v, inv = eng.pass1(dev(a), dev(b), ws, ov, precision=precision)
    u, v, inv = u[0].cpu().numpy(), v[0].cpu().numpy(), inv[0].cpu().numpy().astype(bool)
    const = pass1_constant(a, b, ws, ov)           # flat maps (saturated / black blocks): every cell ties
    tie = exact_tie_windows(a, b, ws, ov) | const
    err = np.maximum(np.abs(u - ru), np.abs(v - rv))
    flips = inv != rmask
    n_tie = int((tie & ~const).sum())
    print(f"  {precision} precision {what} (ws {ws}): max |d| {err[~tie].max() if (~tie).any() else 0:.2e} px over "
          f"{err.size - int(tie.sum())} windows, mask flips {int((flips & ~tie).sum())}, exact float64 ties {n_tie}, "
          f"constant-input windows {int(const.sum())} (of which differing: {int((const & (flips | (err > TOL_REF))).sum())})")
    assert n_tie <= 0.01 * tie.size, (what, n_tie)          # genuine ties are rare (the same 1 % cap as elsewhere)
    assert not (flips & ~tie).any(), (what, np.argwhere(flips & ~tie)[:5].tolist())
    assert err[~tie].max() <= TOL_REF, (what, float(err[~tie].max()), np.argwhere((err > TOL_REF) & ~tie)[:5].tolist())


@pytest.mark.parametrize("precision", ["reference", "exact"])
def test_pass1_reference_precision_goldens(eng, golden, precision):
    """Every pass-1 golden (tile sizes 8..128, generic sizes 6..256, black / saturated blocks) at the
    reference's own precision, and at the library's default ("exact": exact sums for 64x64 windows, the float64 kernels
    for the other sizes): <= 1e-9 px and identical validity masks, no excuse set."""
    g = golden("g3_pass1")
    for name in g["names"]:
        ws, ov = (int(t) for t in g[name + "_cfg"])
        check_reference_precision(eng, g[name + "_a"], g[name + "_b"], ws, ov, g[name + "_u"], g[name + "_v"],
                                  g[name + "_mask"], name, precision)
    g = golden("g7_generic")
    for name in g["p1_names"]:
        ws, ov = (int(t) for t in g[name + "_cfg"])
        check_reference_precision(eng, g[name + "_a"], g[name + "_b"], ws, ov, g[name + "_u"], g[name + "_v"],
                                  g[name + "_mask"], name, precision)        # (incl. the odd size ws33)
    g = golden("g4_multipass")
    for name in g["names"]:
        ws, ov, _ = (int(t) for t in g[name + "_cfg"])
        check_reference_precision(eng, g[name + "_a"], g[name + "_b"], ws, ov, g[name + "_DWS_p0_u"],
                                  g[name + "_DWS_p0_v"], g[name + "_DWS_p0_val"], name + " p0", precision)


@pytest.mark.parametrize("ws,H,W,precision", [(64, 2048, 2048, "reference"), (64, 2048, 2048, "exact"), (32, 1024, 1536, "reference")])
def test_reference_precision_vs_oracle_large(eng, ws, H, W, precision):
    """A full-size frame (configs[1] geometry for ws = 64) against the float64 oracle."""
    from torchpiv_amd import synth
    a, b = synth.make_pair(H, W, 5000 + ws, kind="wavy", noise=2.0)
    ou, ov_, _, _, om = O.pass1(a.numpy(), b.numpy(), ws, ws // 2, validate=True)
    check_reference_precision(eng, a.numpy(), b.numpy(), ws, ws // 2, ou, ov_, om, f"{H}x{W}", precision)


def test_reference_precision_errors(eng):
    a = torch.zeros(64, 64, dtype=torch.uint8).cuda()
    with pytest.raises(KeyError):
        eng.pass1(a, a, 32, 16, precision="double")
    with pytest.raises(KeyError):
        eng.Plan(64, 64, 32, 16, precision="quad")


@pytest.mark.parametrize("mode", ["DWS", "CWS"])
def test_shifted_128_windows(eng, golden, mode):
    """256/128 -> 128/64: the shifted 128x128 pass from the REFERENCE's pass-1 fields (strict, per pass),
    then the whole plan with the propagation rule."""
    g = golden("g8_round2")
    name = "big256x2"
    ws, ov, n_pass = (int(t) for t in g[name + "_cfg"])
    a, b = g[name + "_a"], g[name + "_b"]
    H, W = a.shape
    xc, yc = eng.coordinates_1d(H, W, ws, ov)
    w, o = ws // 2, ov // 2
    xf, yf = eng.coordinates_1d(H, W, w, o)
    Ay, Ax = dev(eng.spline_matrix(yc, yf)), dev(eng.spline_matrix(xc, xf))
    u0, v0, u2, v2 = eng.predict(mode, Ay, Ax, dev(g[f"{name}_{mode}_p0_u"])[None], dev(g[f"{name}_{mode}_p0_v"])[None],
                                 dev(g[f"{name}_{mode}_p0_val"].astype(np.uint8))[None])
    u, v, inv = eng.iterate(mode, dev(a), dev(b), w, o, u0, v0, u2, v2)
    aa, bb = staged_windows(a, b, H, W, w, o, mode, u2, v2)
    nr, nc = O.field_shape((H, W), w, o)
    e, f = check_fields(u[0], v[0], inv[0], g[f"{name}_{mode}_p1_u"], g[f"{name}_{mode}_p1_v"], g[f"{name}_{mode}_p1_val"],
                        f"{name} {mode} pass 1", max_flip_frac=0.0, max_bad_frac=0.0,
                        excused=fp32_noise_excuse(aa, bb, nr, nc), constant=constant_windows(aa, bb, nr, nc), cap=0.03)
    print(f"shifted 128 {mode}: max err {e:.2e} px, mask flips {f}")
    # staged windows of the shifted pass, bit-exact (generic kernel)
    _, _, _, win, _ = eng.debug_pass(mode, dev(a), dev(b), w, o, u2, v2)
    assert np.array_equal(win[0, :, 0].cpu().numpy(), aa.astype(np.float32))
    assert np.array_equal(win[0, :, 1].cpu().numpy(), bb.astype(np.float32))
    for precision in ("reference", "f64", "fast", "exact"):       # ("exact": a 66-pixel first pass -- generic candidate kernel)
        cascade_check(eng, g, name, mode, precision, [(ws, ov), (w, o)], max_differing=[2, 2])       # observed: 0, 0


def test_function_seam_two_streams(eng):
    """tpiv_pass1 / tpiv_iter keep no library state (caller-provided work buffers): the same call on
    two streams at once gives the results of the serial calls."""
    from torchpiv_amd import synth
    pairs = [synth.make_pair(512, 640, 70 + i, kind="wavy", noise=2.0) for i in range(2)]
    serial = [eng.pass1(a.cuda(), b.cuda(), 32, 16) for a, b in pairs]
    torch.cuda.synchronize()
    streams = [torch.cuda.Stream() for _ in pairs]
    outs = []
    for rep in range(4):
        outs = []
        for (a, b), s in zip(pairs, streams):
            with torch.cuda.stream(s):
                outs.append(eng.pass1(a.cuda(), b.cuda(), 32, 16))
        torch.cuda.synchronize()
        for got, want in zip(outs, serial):
            assert all(torch.equal(x, y) for x, y in zip(got, want))


def test_plan_out_validation(eng):
    plan = eng.Plan(128, 128, 32, 16, n_pass=1, max_batch=2)
    a = torch.zeros(2, 128, 128, dtype=torch.uint8).cuda()
    nr, nc = plan.out_shape
    good = (torch.empty(2, nr, nc, dtype=torch.float64).cuda(), torch.empty(2, nr, nc, dtype=torch.float64).cuda(),
            torch.empty(2, nr, nc, dtype=torch.uint8).cuda())
    plan.run(a, a, out=good)
    with pytest.raises(TypeError):
        plan.run(a, a, out=(good[0].float(), good[1], good[2]))
    with pytest.raises(ValueError):
        plan.run(a, a, out=(good[0][:1], good[1], good[2]))
    with pytest.raises(ValueError):
        plan.run(a, a, out=(good[0].cpu(), good[1], good[2]))
    with pytest.raises(ValueError):
        plan.run(torch.zeros(3, 128, 128, dtype=torch.uint8).cuda(), torch.zeros(3, 128, 128, dtype=torch.uint8).cuda())
    plan.close()


# ---------------------------------------------------------------------------------------------
# post-validation on the device (tpiv_postval) against the host SciPy path (= the reference's, pinned by
# tests/test_host_logic.py::test_post_validation_matches_reference on golden g6 pv_*)
# ---------------------------------------------------------------------------------------------
def _host_postval(u, v, val):
    from torchpiv_amd import backend as Bk
    import contextlib, io
    with contextlib.redirect_stdout(io.StringIO()):
        return Bk.post_validate(u.copy(), v.copy(), val)


def _random_masks(rng, nr, nc):
    """Hole patterns: isolated cells (co-circular diamonds), straight runs, blobs, border cells, whole
    border edges, and the two drop cases (no hole; too many)."""
    cases = []
    m = np.zeros((nr, nc), bool)
    cases.append(("none", m.copy()))
    m = rng.random((nr, nc)) < 0.02
    cases.append(("isolated", m))
    m = np.zeros((nr, nc), bool)
    m[3, 2:7] = True
    m[6:11, 9] = True
    m[12, 4:6] = True
    cases.append(("runs only", m))                                    # device-complete
    m = np.zeros((nr, nc), bool)
    m[4:7, 5:9] = True
    m[10, 10] = m[10, 12] = True
    cases.append(("blob", m))
    m = np.zeros((nr, nc), bool)
    m[0, :] = True
    m[5, 0] = m[nr - 1, 3:6] = m[2:4, nc - 1] = True
    m[0, 0] = m[nr - 1, nc - 1] = True
    cases.append(("edges", m))
    m = rng.random((nr, nc)) < 0.3
    cases.append(("too many", m))
    m = np.zeros((nr, nc), bool)
    m[1:nr - 1, 1] = True
    m[1, 3:8] = True
    cases.append(("runs next to the border", m))
    m = np.ones((nr, nc), bool)
    cases.append(("all invalid", m))
    m = np.zeros((nr, nc), bool)
    m[:, 0] = m[:, nc - 1] = True
    m[0, :] = m[nr - 1, :] = True
    cases.append(("whole frame border", m))
    # groups whose every cell has a triangulation-independent fill (round 4: postval_rules.inc): L, T, S shapes, a long L,
    # a thick bar's corner cells stay GENERAL (co-circular), so that one is listed apart
    m = np.zeros((nr, nc), bool)
    m[2, 3] = m[3, 3] = m[3, 4] = True                       # L3
    m[2, 9:12] = m[3, 10] = True                             # T4
    m[7, 3] = m[7, 4] = m[8, 4] = m[8, 5] = True             # S4
    m[10:13, 12] = m[12, 13] = True                          # L4
    m[12, 17:19] = m[13, 18] = m[14, 18] = True              # L4, other orientation
    cases.append(("L, T and S shaped groups", m))            # device-complete
    m = np.zeros((nr, nc), bool)
    m[4:6, 5:9] = True                                       # 2 x 4 bar: co-circular octagons, stays with Qhull
    m[10, 3] = m[11, 3] = m[11, 4] = True
    cases.append(("thick bar + L", m))
    for k, dens in enumerate((0.04, 0.06, 0.08, 0.10, 0.12, 0.08)):      # clustered random holes: whatever groups come up
        m = rng.random((nr, nc)) < dens
        m |= np.roll(m, 1, axis=k % 2) & (rng.random((nr, nc)) < 0.5)
        cases.append((f"random clustered {k}", m))
    return cases


def test_postval_device_vs_host(eng, golden):
    """tpiv_postval on crafted masks: border interpolation bit-compatible with np.interp, the census
    reproduces both drop decisions, device fills equal the Delaunay fill, classes are right."""
    rng = np.random.default_rng(3)
    nr, nc = 17, 23
    names, us, vs, ms = [], [], [], []
    for name, m in _random_masks(rng, nr, nc):
        names.append(name)
        us.append(rng.standard_normal((nr, nc)) * 4 + 2)
        vs.append(rng.standard_normal((nr, nc)) * 3 - 1)
        ms.append(m)
    g = golden("g6_kats")          # the reference's own KAT: NaN pattern of pv_in
    pv = g["pv_in"]
    U = torch.from_numpy(np.stack(us)).cuda()
    V = torch.from_numpy(np.stack(vs)).cuda()
    M = torch.from_numpy(np.stack(ms).astype(np.uint8)).cuda()
    cls, counts = eng.postval(U, V, M)
    cls, counts, Ud, Vd = cls.cpu().numpy(), counts.cpu().numpy(), U.cpu().numpy(), V.cpu().numpy()
    from torchpiv_amd import backend as Bk
    for k, name in enumerate(names):
        u0, v0, m = us[k].copy(), vs[k].copy(), ms[k]
        u0[m] = np.nan
        v0[m] = np.nan
        ub, vb = Bk.interpolate_boarders(u0.copy()), Bk.interpolate_boarders(v0.copy())
        hole = np.isnan(ub)
        assert np.array_equal(hole, (cls[k] >= 1) & (cls[k] <= 4)), name
        ring, _ = Bk.getPixelsForInterp(ub)
        assert np.array_equal(ring, cls[k] == 5), name
        assert counts[k, 0] == hole.sum() and counts[k, 1] == ring.sum(), name
        # border interpolation: np.interp arithmetic, bit for bit, on every cell the device did not fill
        untouched = ~hole
        assert np.array_equal(Ud[k][untouched], ub[untouched]) and np.array_equal(Vd[k][untouched], vb[untouched]), name
        hu, hv = _host_postval(us[k], vs[k], m)
        dropped = ring.sum() == 0 or 4 * ring.sum() >= nr * nc
        if hu is None:
            # the census reproduces the drop, or Qhull refused the ring (a host-fallback pair by class)
            assert dropped or (counts[k, 2] + counts[k, 3]) > 0, name
            continue
        assert not dropped, name
        filled = cls[k] == 2
        assert np.allclose(Ud[k][filled], hu[filled], rtol=0, atol=1e-12), name
        assert np.allclose(Vd[k][filled], hv[filled], rtol=0, atol=1e-12), name
        if counts[k, 2] + counts[k, 3] == 0:
            assert np.allclose(Ud[k], hu, rtol=0, atol=1e-12, equal_nan=True), name
        if name == "L, T and S shaped groups":
            assert counts[k, 2] + counts[k, 3] == 0 and filled.sum() == m.sum(), (name, counts[k].tolist())
        if name == "thick bar + L":
            assert counts[k, 3] > 0 and filled.sum() >= 3, (name, counts[k].tolist())
        print(f"  postval '{name}': holes {counts[k, 0]}, ring {counts[k, 1]}, ambiguous {counts[k, 2]}, "
              f"general {counts[k, 3]}, filled on the device {int(filled.sum())}")
    # the reference's KAT (golden g6 pv_*): same NaN pattern through the device path
    m = np.isnan(pv)
    u = np.where(m, 0.0, pv)
    U = torch.from_numpy(u[None].copy()).cuda()
    V = torch.from_numpy(u[None].copy()).cuda()
    cls, counts = eng.postval(U, V, torch.from_numpy(m[None].astype(np.uint8)).cuda())
    got = U[0].cpu().numpy()
    keep = cls[0].cpu().numpy() != 3
    keep &= cls[0].cpu().numpy() != 4
    assert np.allclose(got[keep], g["pv_filled"][keep], rtol=0, atol=1e-12)
    hole = (cls[0].cpu().numpy() >= 1) & (cls[0].cpu().numpy() <= 4)
    assert np.array_equal(got[~hole], g["pv_borders"][~hole])


def test_postval_compact_lists_are_argwhere_lists(eng):
    """tpiv_postval_compact: for every pair that needs the host triangulation the packed ring cells, their values and
    the hole cells are np.argwhere's lists of the class map (row-major: Qhull's insertion order) -- bit for bit; pairs
    that are dropped or complete on the device take no room.  And the generator gives the same tuples when the lists do
    not fit the asynchronous copy (RING_CAP = 1: the synchronous overflow fetch)."""
    rng = np.random.default_rng(8)
    nr, nc = 37, 41
    names, us, vs, ms = [], [], [], []
    for rep in range(3):
        for name, m in _random_masks(rng, nr, nc):
            names.append(name)
            us.append(rng.standard_normal((nr, nc)))
            vs.append(rng.standard_normal((nr, nc)))
            ms.append(m)
    U = torch.from_numpy(np.stack(us)).cuda()
    V = torch.from_numpy(np.stack(vs)).cuda()
    M = torch.from_numpy(np.stack(ms).astype(np.uint8)).cuda()
    cls, counts = eng.postval(U, V, M)
    off, rc, uv, hc = (t.cpu().numpy() for t in eng.postval_compact(U, V, cls, counts))
    cls_h, cnt, Uh, Vh = cls.cpu().numpy(), counts.cpu().numpy().astype(np.int64), U.cpu().numpy(), V.cpu().numpy()
    B = len(names)
    n_need = 0
    for k in range(B):
        need = cnt[k, 1] > 0 and 4 * cnt[k, 1] < nr * nc and cnt[k, 2] + cnt[k, 3] > 0
        r0, r1, h0, h1 = off[0, k], off[0, k + 1], off[1, k], off[1, k + 1]
        if not need:
            assert r0 == r1 and h0 == h1, names[k]
            continue
        n_need += 1
        ring, hole = cls_h[k] == 5, (cls_h[k] >= 1) & (cls_h[k] <= 4)
        assert np.array_equal(rc[r0:r1], np.argwhere(ring)), names[k]
        assert np.array_equal(hc[h0:h1], np.argwhere(hole)), names[k]
        assert np.array_equal(uv[r0:r1, 0], Uh[k][ring]) and np.array_equal(uv[r0:r1, 1], Vh[k][ring]), names[k]
    assert n_need >= 6 and off[0, B] == off[0, B - 1] + (off[0, B] - off[0, B - 1])
    # overflow path of the generator
    import torchpiv_amd as T
    from torchpiv_amd import synth
    A, Bf = synth.make_batch(6, 256, 320, first_index=40, noise=2.0, device="cuda")
    for i in range(6):
        for (y, x) in ((60, 70), (150, 200), (200, 90)):
            A[i, y:y + 40, x:x + 40 + 8 * i] = 0
            Bf[i, y:y + 40, x:x + 40 + 8 * i] = 0
    outs = []
    for cap in (256, 1):
        piv = T.ResidentPIV(A, Bf, 32, 16, multipass=2, multipass_mode="CWS")
        piv.RING_CAP = cap
        outs.append([tuple(np.array(t) for t in r) for r in piv.batched(4)])
        assert piv.stats["host_fallback"] >= 4
        piv.close()
    assert len(outs[0]) == len(outs[1]) >= 4
    for g_, r_ in zip(*outs):
        assert all(np.array_equal(x, y, equal_nan=True) for x, y in zip(g_, r_))


def test_finish_fields_is_the_reference_expression_bit_for_bit(eng):
    """tpiv_finish_fields = B:894-898 on the device: flip along the rows, sign of v, `u * scale / dt * 1000` left to
    right -- numpy's bits for ordinary values, zeros of either sign, subnormals and huge values, for several scales."""
    rng = np.random.default_rng(21)
    u = rng.standard_normal((3, 17, 23)) * 10.0 ** rng.integers(-12, 12, size=(3, 17, 23))
    v = rng.standard_normal((3, 17, 23)) * 10.0 ** rng.integers(-12, 12, size=(3, 17, 23))
    u[0, 0, :4] = [0.0, -0.0, 5e-324, -2.5e-310]
    v[1, 3, :4] = [0.0, -0.0, 1.7e308, -1.7e308]
    for scale, dt in ((1.0, 1), (0.02, 12), (0.013, 7), (3.3e-5, 1000)):
        fu, fv = eng.finish_fields(torch.from_numpy(u).cuda(), torch.from_numpy(v).cuda(), scale, dt)
        with np.errstate(over="ignore"):
            ru = np.flip(u, axis=1) * scale / dt * 1000
            rv = -np.flip(v, axis=1) * scale / dt * 1000
        assert np.array_equal(fu.cpu().numpy().view(np.int64), ru.view(np.int64)), (scale, dt)
        assert np.array_equal(fv.cpu().numpy().view(np.int64), rv.view(np.int64)), (scale, dt)


def test_postval_single_row_or_column_grid(eng):
    """A final grid with ONE row or column (a short frame under a large window): tpiv_postval accepts it and the
    census drops the pair like the reference does (its interpolator has no usable ring: Qhull refuses collinear
    points / no points at all) -- the old check raised EINVAL here."""
    rng = np.random.default_rng(8)
    for shape in ((1, 9), (7, 1)):
        u = torch.from_numpy(rng.standard_normal((2,) + shape)).cuda()
        v = torch.from_numpy(rng.standard_normal((2,) + shape)).cuda()
        m = np.zeros((2,) + shape, np.uint8)
        m[0].flat[3] = 1
        cls, counts = eng.postval(u, v, torch.from_numpy(m).cuda())
        counts = counts.cpu().numpy()
        # the border interpolation fills the hole (every cell is a border cell), so no ring is left: dropped,
        # exactly as the host path decides (post_validate -> None)
        assert counts[:, 1].tolist() == [0, 0], counts
        for k in range(2):
            hu, hv = _host_postval(u[k].cpu().numpy(), v[k].cpu().numpy(), m[k].astype(bool))
            assert hu is None and hv is None


def test_resident_generator_equals_host_post_validation(eng):
    """ResidentPIV (device post-validation + counted host fallbacks) against the same kernels' raw fields
    pushed through the host-only reference path: identical drops, values to 1e-12."""
    import torchpiv_amd as T
    from torchpiv_amd import synth
    H, W = 512, 640
    A, B = [], []
    for i in range(6):
        a, b = synth.make_pair(H, W, 300 + i, kind=("wavy", "vortex", "shear")[i % 3], noise=(0.0, 3.0, 8.0)[i % 3])
        if i in (1, 4):
            a[100:140, 200:330] = 0            # a dead patch: a blob of invalid vectors in pass 2
            b[100:140, 200:330] = 0
        if i == 5:
            a[:, :] = 7                        # constant frame: everything invalid -> "too many" / no ring
        A.append(a)
        B.append(b)
    A, B = torch.stack(A).cuda(), torch.stack(B).cuda()
    piv = T.ResidentPIV(A, B, 32, 16, multipass=2, multipass_mode="CWS", dt=2, scale=0.5)       # (default precision: "exact" = "f64" for 32x32 windows)
    res = {i: (x, y, u, v) for i, x, y, u, v in piv.batched(4)}
    # (the same precision as the generator's default: pair 5's windows are flat maps -- one frame constant --, whose arg-max is
    #  rounding noise of whichever float64 kernel instance transforms them; this test is about the post-validation)
    plan = eng.Plan(H, W, 32, 16, n_pass=2, mode="CWS", max_batch=6)
    u, v, inv = plan.run(A, B)
    u, v, inv = u.cpu().numpy(), v.cpu().numpy(), inv.cpu().numpy().astype(bool)
    n_kept = 0
    for i in range(6):
        hu, hv = _host_postval(u[i], v[i], inv[i])
        if hu is None:
            assert i not in res, i
            continue
        n_kept += 1
        assert i in res, i
        x, y, gu, gv = res[i]
        wu = np.flip(hu, axis=0) * 0.5 / 2 * 1000
        wv = -np.flip(hv, axis=0) * 0.5 / 2 * 1000
        assert np.allclose(gu, wu, rtol=0, atol=1e-9, equal_nan=True) and np.allclose(gv, wv, rtol=0, atol=1e-9, equal_nan=True), i
    st = piv.stats
    print("  post-validation stats:", st)
    assert st["pairs"] == 6 and st["device_complete"] + st["host_fallback"] - st["dropped_by_qhull"] == n_kept
    assert list(piv()) and len(list(piv())) == n_kept
    plan.close()


def test_generator_with_fill_workers_equals_the_in_process_path(eng, monkeypatch):
    """fill_workers > 0 (round 5: own worker processes, jobs handed out and answers collected one pipeline step apart) yields
    the bits of the in-process path: over several launches, with isolated holes and wide ones, with every batch forced through
    the on-the-spot exchange of heavy messages, after a generator that was abandoned with answers still in the pipes, and
    with the number of workers changed in between."""
    import torchpiv_amd as T
    from torchpiv_amd import synth, _qhull
    H, W = 512, 640
    rng = np.random.default_rng(5)
    A, B = [], []
    for i in range(10):
        a, b = synth.make_pair(H, W, 700 + i, kind=("wavy", "vortex")[i % 2], noise=2.0)
        for _ in range(3):                             # isolated dead windows (co-circular holes: Qhull's diagonal)
            y, x = int(rng.integers(40, H - 80)), int(rng.integers(40, W - 80))
            a[y:y + 20, x:x + 20] = 0
            b[y:y + 20, x:x + 20] = 0
        if i % 4 == 1:                                 # and a blob (a wide hole: the general path)
            a[300:340, 100:230] = 0
            b[300:340, 100:230] = 0
        A.append(a)
        B.append(b)
    A, B = torch.stack(A).cuda(), torch.stack(B).cuda()

    def run(workers, batch):
        piv = T.ResidentPIV(A, B, 32, 16, multipass=2, multipass_mode="CWS")
        piv.fill_workers = workers
        try:
            out = {i: (u.copy(), v.copy()) for i, x, y, u, v in piv.batched(batch)}
            return out, dict(piv.stats), piv
        except Exception:
            piv.close()
            raise

    want, st0, p0 = run(0, 3)
    p0.close()
    assert st0["host_fallback"] >= 8 and st0["host_fallback_wide_holes"] >= 2 and len(want) == 10
    got, st1, piv = run(2, 3)
    try:
        def same(res):
            assert sorted(res) == sorted(want)
            for i in want:
                assert np.array_equal(res[i][0], want[i][0], equal_nan=True) and np.array_equal(res[i][1], want[i][1], equal_nan=True), i
        same(got)
        assert {k: st1[k] for k in st0} == st0
        # abandoned after the first pair: answers of later launches stay in the pipes; the next run must not read them as its own
        g = piv.batched(2)
        next(g)
        g.close()
        same({i: (u.copy(), v.copy()) for i, x, y, u, v in piv.batched(4)})
        # every message "heavy": exchanged on the spot, job by job
        monkeypatch.setattr(_qhull.FillWorkers, "LIGHT", 0)
        same({i: (u.copy(), v.copy()) for i, x, y, u, v in piv.batched(5)})
        monkeypatch.undo()
        piv.fill_workers = 3                           # another number of workers: the old ones go, new ones start
        same({i: (u.copy(), v.copy()) for i, x, y, u, v in piv.batched(10)})
    finally:
        piv.close()


@pytest.mark.parametrize("ws", [16, 32, 64])
def test_fast_staging_close_to_reference_order(eng, ws):
    """precision="fast" forms the CWS sample as row lerps + a column lerp with the reference's float32
    weights: the staged windows differ from the bit-exact (reference-order) ones by float32 rounding only,
    and wavefronts that hold an integral row shift (the "nearest sample" quirk) are bit-exact."""
    rng = np.random.default_rng(900 + ws)
    H, W, ov = 5 * ws + 8, 6 * ws + 4, ws // 2
    fa = rng.integers(0, 256, size=(H, W)).astype(np.uint8)
    fb = rng.integers(0, 256, size=(H, W)).astype(np.uint8)
    nr, nc = O.field_shape((H, W), ws, ov)
    n = nr * nc
    vx = rng.uniform(-0.4 * ws, 0.4 * ws, n).astype(np.float32)
    vy = rng.uniform(-0.4 * ws, 0.4 * ws, n).astype(np.float32)
    vy[::9] = np.rint(vy[::9])
    vxd = torch.from_numpy(vx.astype(np.float64)).cuda().view(1, nr, nc)
    vyd = torch.from_numpy(vy.astype(np.float64)).cuda().view(1, nr, nc)
    _, _, _, w_ref, _ = eng.debug_pass("CWS", dev(fa), dev(fb), ws, ov, vxd, vyd, precision="reference")
    _, _, _, w_fast, _ = eng.debug_pass("CWS", dev(fa), dev(fb), ws, ov, vxd, vyd, precision="fast")
    d = (w_ref - w_fast).abs()
    print(f"  ws {ws}: fast vs reference-order staging: max |d| {d.max().item():.2e} grey levels, "
          f"{(d > 0).float().mean().item():.3f} of the samples differ")
    assert d.max().item() <= 1e-4            # a few float32 ulp of 255
    integral = torch.from_numpy(vy == np.rint(vy)).cuda()
    assert torch.equal(w_ref[0][integral], w_fast[0][integral])


def test_bmp_unpack_on_device_equals_host_decode(eng, tmp_path):
    """tpiv_bmp_unpack against io.decode_bmp_gray, byte for byte: 8-bit grey-ramp and arbitrary palettes,
    24- and 32-bit colour, bottom-up and top-down rows, widths that need row padding."""
    import struct
    from PIL import Image
    from torchpiv_amd import io as pio
    rng = np.random.default_rng(12)
    H, W = 37, 50                                     # W % 4 != 0: padded rows
    files = []
    gray = rng.integers(0, 256, size=(H, W)).astype(np.uint8)
    Image.fromarray(gray, "L").save(tmp_path / "g8.bmp")
    files.append("g8.bmp")
    rgb = rng.integers(0, 256, size=(H, W, 3)).astype(np.uint8)
    Image.fromarray(rgb, "RGB").save(tmp_path / "c24.bmp")
    files.append("c24.bmp")
    pal = Image.fromarray(gray, "L").convert("P", palette=Image.ADAPTIVE, colors=200)
    pal.putpalette(list(rng.integers(0, 256, size=768).astype(np.uint8)))
    pal.save(tmp_path / "p8.bmp")
    files.append("p8.bmp")
    # top-down 32-bit file written by hand (negative height)
    bgra = rng.integers(0, 256, size=(H, W, 4)).astype(np.uint8)
    hdr = b"BM" + struct.pack("<IHHI", 54 + H * W * 4, 0, 0, 54) + struct.pack("<IiiHHIIiiII", 40, W, -H, 1, 32, 0, H * W * 4,
                                                                                2835, 2835, 0, 0)
    (tmp_path / "t32.bmp").write_bytes(hdr + bgra.tobytes())
    files.append("t32.bmp")
    cap = max((tmp_path / f).stat().st_size for f in files)
    stage = torch.zeros(len(files), cap, dtype=torch.uint8).pin_memory()
    desc, luts, want = [], [], []
    for k, f in enumerate(files):
        lay = pio.stage_raw(str(tmp_path / f), stage[k].numpy(), H, W)
        assert lay is not None, f
        desc.append([k * cap, lay[0], lay[1], lay[2], lay[3], 0])
        luts.append(lay[4])
        want.append(pio.decode_bmp_gray((tmp_path / f).read_bytes()))
    out = eng.bmp_unpack(stage.cuda().view(-1), torch.tensor(desc, dtype=torch.int64).cuda(),
                         torch.from_numpy(np.stack(luts)).cuda(), H, W).cpu().numpy()
    for k, f in enumerate(files):
        assert np.array_equal(out[k], want[k]), f
    assert np.array_equal(want[0], gray)
    # a PNG goes through the host decoder and travels as headerless pixels
    Image.fromarray(gray, "L").save(tmp_path / "g.png")
    lay = pio.stage_raw(str(tmp_path / "g.png"), stage[0].numpy(), H, W)
    assert lay[:4] == (0, W, 1, 0)
    out = eng.bmp_unpack(stage.cuda().view(-1), torch.tensor([[0, lay[0], lay[1], lay[2], lay[3], 0]], dtype=torch.int64).cuda(),
                         torch.from_numpy(lay[4][None].copy()).cuda(), H, W).cpu().numpy()
    assert np.array_equal(out[0], gray)


@pytest.mark.parametrize("mode", ["DWS", "CWS"])
def test_odd_window_in_a_shifted_pass(eng, golden, mode):
    """66/33 -> 33/16: an ODD window size in a shifted pass (the reference's 33 x 32 correlation map, B:255
    irfft2 without `s`; flat-index rules with k = 32 columns and d = 33 rows), per pass from the reference's
    fields and as a whole plan."""
    g = golden("g8_round2")
    name = "odd66x2"
    ws, ov, n_pass = (int(t) for t in g[name + "_cfg"])
    a, b = g[name + "_a"], g[name + "_b"]
    H, W = a.shape
    xc, yc = eng.coordinates_1d(H, W, ws, ov)
    w, o = ws // 2, ov // 2
    xf, yf = eng.coordinates_1d(H, W, w, o)
    Ay, Ax = dev(eng.spline_matrix(yc, yf)), dev(eng.spline_matrix(xc, xf))
    u0, v0, u2, v2 = eng.predict(mode, Ay, Ax, dev(g[f"{name}_{mode}_p0_u"])[None], dev(g[f"{name}_{mode}_p0_v"])[None],
                                 dev(g[f"{name}_{mode}_p0_val"].astype(np.uint8))[None])
    u, v, inv = eng.iterate(mode, dev(a), dev(b), w, o, u0, v0, u2, v2)
    aa, bb = staged_windows(a, b, H, W, w, o, mode, u2, v2)
    nr, nc = O.field_shape((H, W), w, o)
    e, f = check_fields(u[0], v[0], inv[0], g[f"{name}_{mode}_p1_u"], g[f"{name}_{mode}_p1_v"], g[f"{name}_{mode}_p1_val"],
                        f"{name} {mode} pass 1", max_flip_frac=0.0, max_bad_frac=0.0,
                        excused=fp32_noise_excuse(aa, bb, nr, nc), constant=constant_windows(aa, bb, nr, nc))
    print(f"odd window 33 {mode}: max err {e:.2e} px, mask flips {f}")
    for precision in ("reference", "f64", "fast", "exact"):       # ("exact": a 66-pixel first pass -- generic candidate kernel)
        cascade_check(eng, g, name, mode, precision, [(ws, ov), (w, o)], max_differing=[2, 2])       # observed: 0, 0


@pytest.mark.parametrize("precision", ["reference", "f64", "fast", "exact"])
@pytest.mark.parametrize("mode", ["DWS", "CWS"])
def test_other_scales_and_zero_overlap(eng, golden, mode, precision):
    """Whole plans at multipass_scale 1.5 (64/32 -> 42/21 -> 28/14: generic sizes in shifted passes), 4.0
    (64/32 -> 16/8) and with zero overlap (32/0 -> 16/0) against the reference's fields of every pass
    (golden g11), by the threshold-free cascade rule."""
    g = golden("g11_scales")
    for name in g["names"]:
        geo = [tuple(int(t) for t in row) for row in g[name + "_geo"]]
        cascade_check(eng, g, str(name), mode, precision, geo, scale=float(g[name + "_scale"][0]),
                      max_differing=[2] * len(geo))                                  # observed: 0 in every pass


def test_cws_fast_iteration_golden(eng, golden):
    """piv_iteration_CWS_Fast (B:599-675; bicubic resampling of each window inside itself) through the
    drop-in class, against the reference's own output and with the oracle's staged windows for the noise band."""
    import torchpiv_amd as T
    g = golden("g10_cws_fast")
    for name in g["names"]:
        ws, ov = (int(t) for t in g[name + "_cfg"])
        a, b = g[name + "_a"], g[name + "_b"]
        H, W = a.shape
        w, o = ws // 2, ov // 2
        x, y = O.coordinates(a.shape, ws, ov)
        it = T.piv_iteration_CWS_Fast(a.shape, w, o, "cuda:0")
        u, v, x1, y1, val = it(dev(a), dev(b), x, y, g[name + "_p0_u"].copy(), g[name + "_p0_v"].copy(),
                               g[name + "_p0_val"].copy(), w, o, "cuda:0")
        ref = O.IterCWSFast(a.shape, w, o)(a, b, x, y, g[name + "_p0_u"].copy(), g[name + "_p0_v"].copy(),
                                           g[name + "_p0_val"].copy(), debug=True)
        aa, bb = ref[9], ref[10]
        nr, nc = u.shape
        # the reference correlates mean-normalised float32 windows here: noise band relative to that map
        exc = fp32_noise_excuse(aa, bb, nr, nc, ulps=64.0)
        err = np.maximum(np.abs(u - g[name + "_p1_u"]), np.abs(v - g[name + "_p1_v"]))
        flips = val != g[name + "_p1_val"]
        const = constant_windows(O.windows(a, w, o), O.windows(b, w, o), nr, nc)
        bad = ((err > TOL_PX) | flips) & ~exc & ~const
        print(f"  CWS_Fast {name}: max err {err[~flips & ~exc & ~const].max():.2e} px, flips {int(flips.sum())}, "
              f"in the noise band {int((exc & ~const).sum())}, constant windows {int(const.sum())}, unexplained {int(bad.sum())}")
        assert not bad.any(), (name, np.argwhere(bad)[:5].tolist())
        assert (exc & ~const).mean() <= 0.02
        assert np.array_equal(x1, it.x)
    assert "CWS_Fast" not in T.IterModMap.functions            # unreachable from OfflinePIV, as in the reference
    with pytest.raises(KeyError):
        T.OnlinePIV("x", "no-such-device", "bmp", 32, 16)
    assert T.OnlinePIV("x", "cuda:0", "bmp", 32, 16)._device.type == "cuda"
