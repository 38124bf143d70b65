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


def check_reference_precision(eng, a, b, ws, ov, ru, rv, rmask, what):
    u, v, inv = eng.pass1(dev(a), dev(b), ws, ov, precision="reference")
    u, v, inv = u[0].cpu().numpy(), v[0].cpu().numpy(), inv[0].cpu().numpy().astype(bool)
    const = pass1_constant(a, b, ws, ov)           # flat maps (saturated / black blocks): every cell ties
    tie = exact_tie_windows(a, b, ws, ov) | const
    err = np.maximum(np.abs(u - ru), np.abs(v - rv))
    flips = inv != rmask
    n_tie = int((tie & ~const).sum())
    print(f"  reference precision {what} (ws {ws}): max |d| {err[~tie].max() if (~tie).any() else 0:.2e} px over "
          f"{err.size - int(tie.sum())} windows, mask flips {int((flips & ~tie).sum())}, exact float64 ties {n_tie}, "
          f"constant-input windows {int(const.sum())} (of which differing: {int((const & (flips | (err > TOL_REF))).sum())})")
    assert n_tie <= 0.01 * tie.size, (what, n_tie)          # genuine ties are rare (the same 1 % cap as elsewhere)
    assert not (flips & ~tie).any(), (what, np.argwhere(flips & ~tie)[:5].tolist())
    assert err[~tie].max() <= TOL_REF, (what, float(err[~tie].max()), np.argwhere((err > TOL_REF) & ~tie)[:5].tolist())


def test_pass1_reference_precision_goldens(eng, golden):
    """Every pass-1 golden (tile sizes 8..128, generic sizes 6..256, black / saturated blocks) at the
    reference's own precision: <= 1e-9 px and identical validity masks, no excuse set."""
    g = golden("g3_pass1")
    for name in g["names"]:
        ws, ov = (int(t) for t in g[name + "_cfg"])
        check_reference_precision(eng, g[name + "_a"], g[name + "_b"], ws, ov, g[name + "_u"], g[name + "_v"],
                                  g[name + "_mask"], name)
    g = golden("g7_generic")
    for name in g["p1_names"]:
        ws, ov = (int(t) for t in g[name + "_cfg"])
        if ws % 2:
            continue
        check_reference_precision(eng, g[name + "_a"], g[name + "_b"], ws, ov, g[name + "_u"], g[name + "_v"],
                                  g[name + "_mask"], name)
    g = golden("g4_multipass")
    for name in g["names"]:
        ws, ov, _ = (int(t) for t in g[name + "_cfg"])
        check_reference_precision(eng, g[name + "_a"], g[name + "_b"], ws, ov, g[name + "_DWS_p0_u"],
                                  g[name + "_DWS_p0_v"], g[name + "_DWS_p0_val"], name + " p0")


@pytest.mark.parametrize("ws,H,W", [(64, 2048, 2048), (32, 1024, 1536)])
def test_reference_precision_vs_oracle_large(eng, ws, H, W):
    """A full-size frame (configs[1] geometry for ws = 64) against the float64 oracle."""
    from torchpiv_amd import synth
    a, b = synth.make_pair(H, W, 5000 + ws, kind="wavy", noise=2.0)
    ou, ov_, _, _, om = O.pass1(a.numpy(), b.numpy(), ws, ws // 2, validate=True)
    check_reference_precision(eng, a.numpy(), b.numpy(), ws, ws // 2, ou, ov_, om, f"{H}x{W}")


def test_reference_precision_errors(eng):
    a = torch.zeros(64, 64, dtype=torch.uint8).cuda()
    with pytest.raises(KeyError):
        eng.pass1(a, a, 32, 16, precision="double")
    with pytest.raises(KeyError):
        eng.Plan(64, 64, 32, 16, precision="exact")


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
    for precision in ("reference", "fast"):
        cascade_check(eng, g, name, mode, precision, [(ws, ov), (w, o)],
                      noise_ulps=16.0 if precision == "reference" else 4096.0)


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
