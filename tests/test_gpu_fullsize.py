"""BASELINE.json sizes on the GPU: direct comparison with the oracle on one 2048x2048 pair
(2-pass CWS and DWS, wind 64 / overlap 32) and size-independent properties."""
import numpy as np
import pytest
import torch

from oracle import piv_oracle as O

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def pair():
    from torchpiv_amd import synth
    a, b = synth.make_pair(2048, 2048, 321, kind="wavy", noise=3.0, device="cuda")
    return a, b


def _oracle_fields(an, bn, geo, mode, name):
    """The oracle's fields of every pass in the layout of the goldens (the oracle is pinned to the reference's
    own goldens by tests/test_oracle_golden.py)."""
    g = {name + "_a": an, name + "_b": bn}
    u, v, x, y, val = O.pass1(an, bn, geo[0][0], geo[0][1], validate=True)
    g[f"{name}_{mode}_p0_u"], g[f"{name}_{mode}_p0_v"], g[f"{name}_{mode}_p0_val"] = u, v, val
    for p in range(1, len(geo)):
        u, v, x, y, val = O.ITER[mode](an.shape, geo[p][0], geo[p][1])(an, bn, x, y, u.copy(), v.copy(), val.copy())
        g[f"{name}_{mode}_p{p}_u"], g[f"{name}_{mode}_p{p}_v"], g[f"{name}_{mode}_p{p}_val"] = u.copy(), v.copy(), val.copy()
    return g


@pytest.mark.parametrize("precision", ["reference", "f64", "fast", "exact"])
@pytest.mark.parametrize("mode", ["CWS", "DWS"])
def test_cfg2_against_oracle(pair, mode, precision):
    """configs[1]/[2] geometry at full size (2048^2, 64/32 -> 32/16, 16 129 vectors) against the oracle by the three
    gates of the golden tests (cascade_check): reference chain with a capped 16-ulp band, per-pass isolation
    against the oracle fed with the GPU's own fields, counted drift at the float32 / fast-order precisions."""
    from torchpiv_amd import engine
    from test_gpu_parity import cascade_check
    a, b = pair
    geo = [(64, 32), (32, 16)]
    g = _oracle_fields(a.cpu().numpy(), b.cpu().numpy(), geo, mode, "cfg2")
    counts = cascade_check(engine, g, "cfg2", mode, precision, geo, max_differing=[2, 8])     # observed: 0, 0
    assert counts[-1][-1] == 127 * 127


@pytest.mark.parametrize("precision", ["reference", "f64", "fast", "exact"])
def test_cfg4_large_windows_against_oracle(pair, precision):
    """configs[4] geometry (2048^2, 128/64 -> 64/32, 2-pass CWS; pass 1 runs the two-threads-per-line 128x128
    kernel, its float64 form at reference precision, or -- at the default "exact" -- the 128x128 locating pass + exact
    integer sums) against the oracle, same rule."""
    from torchpiv_amd import engine
    from test_gpu_parity import cascade_check
    a, b = pair
    geo = [(128, 64), (64, 32)]
    g = _oracle_fields(a.cpu().numpy(), b.cpu().numpy(), geo, "CWS", "cfg4")
    counts = cascade_check(engine, g, "cfg4", "CWS", precision, geo, max_differing=[2, 4])        # observed: 0, 0
    assert counts[0][-1] == 31 * 31 and counts[-1][-1] == 63 * 63


@pytest.mark.parametrize("mode", ["CWS", "DWS"])
def test_cfg3_geometry_against_oracle(mode):
    """configs[3]'s pass schedule (32/16 -> 16/8 -> 8/4, three passes: the 16x16 patch-load path, the
    lane-per-window 8x8 kernel and the matrix-core predictor on a 255 x 383 grid) on a 1024 x 1536 pair
    against the oracle, same rule."""
    from torchpiv_amd import engine, synth
    from test_gpu_parity import cascade_check
    a, b = synth.make_pair(1024, 1536, 654, kind="vortex", noise=2.0)
    geo = [(32, 16), (16, 8), (8, 4)]
    g = _oracle_fields(a.numpy(), b.numpy(), geo, mode, "cfg3")
    for precision in ("reference", "f64", "fast", "exact"):
        # absolute caps on the vectors that differ from the oracle's chain at all (observed: 0, 0 and -- DWS, 8x8 pass --
        # 5 of 97 665, every one inside the 16-ulp band)
        counts = cascade_check(engine, g, "cfg3", mode, precision, geo, max_differing=[2, 4, 16])
        assert counts[-1][-1] == 255 * 383


FULLSIZE = __import__("os").environ.get("TPIV_FULLSIZE") == "1"


def test_cfg3_full_size_against_oracle():
    """configs[3] LITERALLY: one 4096 x 4096 pair, 32/16 -> 16/8 -> 8/4, 3-pass CWS, 1 046 529 final vectors, against
    the oracle by the same three gates at the default precision ("exact": exact sums in the 32x32 first pass);
    TPIV_FULLSIZE=1 adds the float64-FFT and the all-float32 runs (further passes over the same oracle fields)."""
    from torchpiv_amd import engine, synth
    from test_gpu_parity import cascade_check
    a, b = synth.make_pair(4096, 4096, 987, kind="wavy", noise=2.0)
    geo = [(32, 16), (16, 8), (8, 4)]
    g = _oracle_fields(a.numpy(), b.numpy(), geo, "CWS", "cfg3full")
    for precision in (("exact", "f64", "fast") if FULLSIZE else ("exact",)):
        counts = cascade_check(engine, g, "cfg3full", "CWS", precision, geo, max_differing=[8, 32, 128])     # observed: 0, 0, 0
        assert counts[-1][-1] == 1023 * 1023


def _cfg2_stream(n, distinct, picks):
    """configs[2] on one GPU: an n-pair stream of 2048 x 2048 frames, 64/32 -> 32/16 2-pass DWS at the default precision,
    in 500-pair shards.  Size-independent properties over the WHOLE stream -- a pair's field does not depend on the shard
    it sits in (the stream re-cut into 256-pair launches and the picked pairs re-run alone give the same bits), two runs
    give the same bits -- and a direct comparison of the picked pairs with the oracle."""
    from torchpiv_amd import engine, synth
    H = W = 2048
    # (rendering a pair takes ~0.1 s: `distinct` pairs laid out over the stream in a shuffled order)
    A0, B0 = synth.make_batch(distinct, H, W, first_index=5000, noise=2.0, device="cuda")
    print(f"  cfg2 stream: {distinct} distinct pairs rendered", flush=True)
    order = torch.from_numpy(np.random.default_rng(4).permutation(n) % distinct).cuda()
    A, B = A0[order], B0[order]
    del A0, B0
    plan = engine.Plan(H, W, 64, 32, n_pass=2, mode="DWS", max_batch=500, precision="exact")

    def stream(step):
        us, vs, ivs = [], [], []
        for s0 in range(0, n, step):
            u, v, inv = plan.run(A[s0:s0 + step], B[s0:s0 + step])
            us.append(u.clone()), vs.append(v.clone()), ivs.append(inv.clone())
        return torch.cat(us), torch.cat(vs), torch.cat(ivs)

    u5, v5, i5 = stream(500)
    print("  cfg2 stream: 500-pair shards done", flush=True)
    u2, v2, i2 = stream(256)
    assert torch.equal(u5, u2) and torch.equal(v5, v2) and torch.equal(i5, i2)
    sums = (u5.view(torch.int64).sum(dim=(1, 2)) ^ v5.view(torch.int64).sum(dim=(1, 2))).cpu().numpy()
    print(f"  cfg2 stream: {n} pairs, shards of 500 and 256 bit-identical; checksum of checksums {int(np.sum(sums * (np.arange(n, dtype=np.int64) * 2 + 1))) & 0xffffffffffffffff:#x}; "
          f"invalid vectors {int(i5.sum())} of {i5.numel()}")
    u5b, v5b, _ = stream(500)
    assert torch.equal(u5, u5b) and torch.equal(v5, v5b)                       # run to run
    one = engine.Plan(H, W, 64, 32, n_pass=2, mode="DWS", max_batch=1, precision="exact")
    worst = 0.0
    for k in picks:
        u1, v1, i1 = one.run(A[k], B[k])
        assert torch.equal(u1[0], u5[k]) and torch.equal(v1[0], v5[k]) and torch.equal(i1[0], i5[k])
        an, bn = A[k].cpu().numpy(), B[k].cpu().numpy()
        ou, ov_, x, y, oval = O.pass1(an, bn, 64, 32, validate=True)
        ou, ov_, x, y, oval = O.ITER["DWS"](an.shape, 32, 16)(an, bn, x, y, ou, ov_, oval)
        gu, gv, gi = u5[k].cpu().numpy(), v5[k].cpu().numpy(), i5[k].cpu().numpy().astype(bool)
        differ = (np.abs(gu - ou) > 1e-3) | (np.abs(gv - ov_) > 1e-3) | (gi != oval)
        same = ~differ
        worst = max(worst, float(np.maximum(np.abs(gu - ou), np.abs(gv - ov_))[same].max()))
        print(f"  cfg2 stream pair {k}: {int(differ.sum())} of {differ.size} vectors differ from the oracle; max |d| elsewhere "
              f"{float(np.maximum(np.abs(gu - ou), np.abs(gv - ov_))[same].max()):.2e} px")
        assert differ.sum() <= 8                                                   # observed: 0
    print(f"  cfg2 stream: worst |d| over {len(picks)} pairs {worst:.2e} px")
    plan.close()
    one.close()


def test_cfg2_stream():
    """configs[2] trimmed to 1 000 pairs (8 GB of frames; two full shards): shard-size invariance bit for bit over the whole
    stream, run-to-run identity, 4 pairs against the oracle.  The literal 4 000-pair form is the next test."""
    _cfg2_stream(1000, 100, [0, 499, 500, 999])


@pytest.mark.skipif(not FULLSIZE, reason="opt-in (TPIV_FULLSIZE=1): the literal 4000-pair stream (33 GB of frames); "
                                         "output kept in profiles/r03/fullsize_cfg2.txt; the 1000-pair form runs by default")
def test_cfg2_stream_literally():
    """configs[2] LITERALLY on one GPU: 4000 pairs, 200 distinct renderings, 12 pairs against the oracle."""
    _cfg2_stream(4000, 200, [0, 1, 499, 500, 777, 1234, 1999, 2000, 2718, 3141, 3998, 3999])


def test_large_windows_invariances():
    """128x128 pass 1: bit-identical fields wherever a pair sits in a batch, and cropping both frames
    by one grid step (64 px) moves the field by exactly one cell."""
    from torchpiv_amd import engine, synth
    a, b = synth.make_pair(1024 + 64, 1024 + 64, 78, kind="shear", noise=2.0, device="cuda")
    a2, b2 = synth.make_pair(1024 + 64, 1024 + 64, 79, kind="vortex", noise=2.0, device="cuda")
    A = torch.stack([a, a2, a, a2, a2])
    B = torch.stack([b, b2, b, b2, b2])
    u, v, inv = engine.pass1(A, B, 128, 64)
    assert torch.equal(u[0], u[2]) and torch.equal(v[0], v[2]) and torch.equal(inv[0], inv[2])
    assert torch.equal(u[1], u[3]) and torch.equal(u[1], u[4]) and torch.equal(v[3], v[4])
    u0, v0, i0 = engine.pass1(a[:1024, :1024].contiguous(), b[:1024, :1024].contiguous(), 128, 64)
    u1, v1, i1 = engine.pass1(a[64:, 64:].contiguous(), b[64:, 64:].contiguous(), 128, 64)
    assert torch.equal(u0[0, 1:, 1:], u1[0, :-1, :-1]) and torch.equal(v0[0, 1:, 1:], v1[0, :-1, :-1])
    assert torch.equal(i0[0, 1:, 1:], i1[0, :-1, :-1])


@pytest.mark.parametrize("precision", ["exact", "fast"])
def test_64x64_cws_pass_with_every_item_on_the_list(precision):
    """The 64x64 CWS pass runs as two launches (round 5): a fast-path-only kernel that sets aside the items with an
    integral shift or a frame-border window, and the full kernel over the list of those.  Extremes of the list: (i)
    byte-identical frames -- the predictor is exactly 0 everywhere, EVERY item is listed -- give a zero field to the float32
    rounding of the second pass (1e-7 px; the first pass is exactly 0 at "exact");
    (ii) mixed batches (identical pair, ordinary pairs) give each pair the bits it has alone, wherever it sits;
    (iii) the ordinary pair agrees with the reference-order kernel (one launch, the per-pixel path in the same kernel)
    to float32 rounding."""
    from torchpiv_amd import engine, synth
    H = W = 1024
    a, b = synth.make_pair(H, W, 91, kind="wavy", noise=2.0, device="cuda")
    a2, b2 = synth.make_pair(H, W, 92, kind="vortex", noise=2.0, device="cuda")
    plan = engine.Plan(H, W, 128, 64, n_pass=2, mode="CWS", max_batch=4, precision=precision)
    assert plan.geometry[1][0] == 64
    u, v, inv = plan.run(torch.stack([a, a]), torch.stack([a, a]))             # (i)
    assert float(u.abs().max()) < 1e-5 and float(v.abs().max()) < 1e-5 and int(inv.sum()) == 0
    u_same, v_same = u[0].clone(), v[0].clone()
    A, B = torch.stack([a, a, a2, a]), torch.stack([b, a, b2, b])               # (ii)
    u, v, inv = plan.run(A, B)
    assert torch.equal(u[0], u[3]) and torch.equal(v[0], v[3]) and torch.equal(inv[0], inv[3])
    assert torch.equal(u[1], u_same) and torch.equal(v[1], v_same)
    u1, v1, inv1 = plan.run(a2, b2)
    assert torch.equal(u1[0], u[2]) and torch.equal(v1[0], v[2]) and torch.equal(inv1[0], inv[2])
    ref = engine.Plan(H, W, 128, 64, n_pass=2, mode="CWS", max_batch=1, precision="reference")    # (iii)
    ur, vr, invr = ref.run(a, b)
    same = (inv[0] == 0) & (invr[0] == 0)
    assert same.float().mean() > 0.9
    d = max(float((u[0] - ur[0])[same].abs().max()), float((v[0] - vr[0])[same].abs().max()))
    print(f"  {precision}: two-launch 64x64 CWS pass against the reference-order kernel: max |d| {d:.2e} px on {int(same.sum())} vectors")
    assert d < 1e-3
    plan.close()
    ref.close()


def test_batch_position_invariance(pair):
    """A pair gives bit-identical fields wherever it sits in a batch (windows are independent)."""
    from torchpiv_amd import engine, synth
    a, b = pair
    a2, b2 = synth.make_pair(2048, 2048, 322, kind="vortex", noise=2.0, device="cuda")
    plan = engine.Plan(2048, 2048, 64, 32, n_pass=2, mode="CWS", max_batch=5)
    A = torch.stack([a, a2, a2, a, a2])
    B = torch.stack([b, b2, b2, b, b2])
    u, v, inv = plan.run(A, B)
    assert torch.equal(u[0], u[3]) and torch.equal(v[0], v[3]) and torch.equal(inv[0], inv[3])
    assert torch.equal(u[1], u[2]) and torch.equal(u[1], u[4])
    u1, v1, inv1 = engine.Plan(2048, 2048, 64, 32, n_pass=2, mode="CWS", max_batch=1).run(a, b)
    assert torch.equal(u1[0], u[0]) and torch.equal(inv1[0], inv[0])
    plan.close()


def test_translation_covariance():
    """Pass 1 depends on window content only: cropping both frames by one grid step (32 px) moves
    the field by exactly one cell."""
    from torchpiv_amd import engine, synth
    a, b = synth.make_pair(1024 + 32, 1024 + 32, 77, kind="shear", noise=2.0, device="cuda")
    u0, v0, i0 = engine.pass1(a[:1024, :1024].contiguous(), b[:1024, :1024].contiguous(), 64, 32)
    u1, v1, i1 = engine.pass1(a[32:, 32:].contiguous(), b[32:, 32:].contiguous(), 64, 32)
    assert torch.equal(u0[0, 1:, 1:], u1[0, :-1, :-1]) and torch.equal(v0[0, 1:, 1:], v1[0, :-1, :-1])
    assert torch.equal(i0[0, 1:, 1:], i1[0, :-1, :-1])


def test_known_displacement_recovered():
    """Synthetic uniform flow (2.3, -1.6) px: the measured field recovers it (sign convention:
    +u = frame b shifted towards +x, +v towards +y) on 4096^2 with three passes down to 8x8 tiles."""
    from torchpiv_amd import engine, synth
    a, b = synth.make_pair(4096, 4096, 9, kind="uniform", noise=1.0, device="cuda")
    plan = engine.Plan(4096, 4096, 32, 16, n_pass=3, mode="CWS", max_batch=1)
    u, v, inv = plan.run(a, b)
    assert u.shape[1:] == (1023, 1023)
    ok = inv[0] == 0
    assert ok.float().mean() > 0.7
    assert abs(u[0][ok].median().item() - 2.3) < 0.05 and abs(v[0][ok].median().item() + 1.6) < 0.05
    plan.close()
