"""The headless runner (ensemble statistics + export formats of the reference's GUI worker) against
goldens made by the reference's OWN PIVWorker.run / save_table / save_binary
(tests/golden/make_golden_stats.py -> g9_stats.npz): the files must come out byte for byte."""
import os

import numpy as np
import pytest


def _pairs(g):
    n = int(g["n_pairs"][0])
    return [tuple(g[f"pair{j}_{k}"] for k in ("x", "y", "u", "v")) for j in range(n)]


def _table(g):
    return {str(k): g[f"table_{j}"] for j, k in enumerate(g["table_keys"])}


def test_statistics_equal_the_reference_bit_for_bit(golden):
    from torchpiv_amd.runner import EnsembleStats
    g = golden("g9_stats")
    pairs = _pairs(g)
    st = EnsembleStats()
    for x, y, u, v in pairs:
        st.add(u, v)
    got = st.table(pairs[-1][0], pairs[-1][1])
    want = _table(g)
    assert list(got) == list(want) and len(got) == 13
    for k in want:
        assert np.array_equal(got[k], want[k]), k
    # fields added out of order (as ranks deliver them) come out the same
    st2 = EnsembleStats()
    for j in (3, 0, 5, 1, 4, 2)[:len(pairs)]:
        st2.add(pairs[j][2], pairs[j][3], index=j)
    assert all(np.array_equal(st2.table(pairs[0][0], pairs[0][1])[k], want[k]) for k in want)


def test_export_files_equal_the_reference_byte_for_byte(tmp_path, golden):
    from torchpiv_amd.runner import EnsembleStats, KEYS_PAIR, save_binary, save_table, uniquify
    g = golden("g9_stats")
    pairs = _pairs(g)
    st = EnsembleStats()
    for opt, tag, save in (("Save all text", "txt", save_table), ("Save all binary", "bin", save_binary)):
        d = str(tmp_path / ("Out_" + tag))
        for x, y, u, v in pairs:
            st.add(u, v)
            save(f"run A_pair.{'txt' if tag == 'txt' else 'npy'}", d, dict(zip(KEYS_PAIR, (x, y, u, v))))
        save_table("run A_statistics.txt", d, _table(g))
        names = sorted(os.listdir(d))
        assert names == [str(n) for n in g[f"{tag}_names"]]
        for j, nm in enumerate(names):
            assert open(os.path.join(d, nm), "rb").read() == g[f"{tag}_file{j}"].tobytes(), (tag, nm)
    assert uniquify(str(tmp_path / "nothing.txt")).endswith("nothing.txt")
    # never overwrites: " (n)" is appended
    p1 = save_table("t.txt", str(tmp_path), {"a": np.arange(3.0)})
    p2 = save_table("t.txt", str(tmp_path), {"a": np.arange(3.0)})
    assert os.path.basename(p1) == "t.txt" and os.path.basename(p2) == "t (1).txt"


@pytest.mark.gpu
def test_device_moments_equal_numpy_bit_for_bit(golden):
    import torch
    from torchpiv_amd import engine
    from torchpiv_amd.runner import EnsembleStats
    g = golden("g9_stats")
    pairs = _pairs(g)
    st = EnsembleStats()
    for x, y, u, v in pairs:
        st.add(u, v)
    host = st.moments()
    dev = st.moments(device="cuda:0")
    for h, d in zip(host, dev):
        assert np.array_equal(h, d)
    rng = np.random.default_rng(4)
    U = rng.standard_normal((257, 31, 45)) * 1e3 + 5
    V = rng.standard_normal((257, 31, 45)) * 1e-3 - 2
    out = engine.ensemble_moments(torch.from_numpy(U).cuda(), torch.from_numpy(V).cuda())
    mu = np.mean(U, axis=0)
    assert np.array_equal(out[0].cpu().numpy(), mu)
    assert np.array_equal(out[2].cpu().numpy(), np.mean((U - mu) ** 2, axis=0))
    assert np.array_equal(out[4].cpu().numpy(), np.mean((U - mu) * (V - np.mean(V, axis=0)), axis=0))


@pytest.mark.gpu
def test_run_folder_matches_generator(tmp_path, golden):
    from PIL import Image
    import torchpiv_amd as T
    from torchpiv_amd.runner import EnsembleStats, run_folder
    g = golden("g9_stats")
    d = tmp_path / "run A"
    d.mkdir()
    for i, (a, b) in enumerate(zip(g["frames_a"], g["frames_b"])):
        Image.fromarray(a, "L").save(d / f"img{8 + i}_a.bmp")
        Image.fromarray(b, "L").save(d / f"img{8 + i}_b.bmp")
    ws, ov, mp_, mode, dt = (int(t) for t in g["kw"])
    kw = dict(wind_size=ws, overlap=ov, multipass=mp_, multipass_mode=("DWS", "CWS")[mode], dt=dt, scale=float(g["scale"][0]))
    out = str(tmp_path / "Out")
    table, n = run_folder(str(d), "cuda:0", "bmp", save_opt="Save all text", save_dir=out, batch_size=4, **kw)
    res = list(T.OfflinePIV(str(d), "cuda:0", "bmp", **kw)())
    assert n == len(res) == int(g["n_pairs"][0])
    st = EnsembleStats()
    for x, y, u, v in res:
        st.add(u, v)
    ref = st.table(res[0][0], res[0][1])
    for k, v in ref.items():
        assert np.array_equal(table[k], v, equal_nan=True), k
    files = sorted(os.listdir(out))
    assert files == [str(nm) for nm in g["txt_names"]]
    hdr = open(os.path.join(out, "run A_statistics.txt")).readline().strip().split(", ")
    assert hdr == list(table.keys())
    # against the reference's own table: the mean field within the parity tolerance of the kernels
    want = _table(g)
    unit = 1000 * kw["scale"] / kw["dt"]
    close = np.isclose(table["Vx[m/s]"] / unit, want["Vx[m/s]"] / unit, rtol=0, atol=1e-3, equal_nan=True)
    # cells downstream of a coin-toss window in ANY pair of the ensemble may differ (tests/test_gpu_api.py)
    from test_gpu_api import explained_region
    region = np.zeros(close.shape, bool)
    for a, b in zip(g["frames_a"], g["frames_b"]):
        region |= explained_region(a, b, ws, ov, mp_, ("DWS", "CWS")[mode])[0]
    print(f"  mean field within 1e-3 px of the reference's: {close.mean():.4f}; unexplained cells "
          f"{int((~close & ~region).sum())}; explained region {region.mean():.2f}")
    # (the ensemble holds pairs with black / saturated blocks, whose coin-toss windows explain much of this small
    #  grid; the region-free statement is the bit-equality with the generator's own tuples above, and the
    #  generator's parity is tests/test_gpu_api.py::test_offline_piv_generator's strict chain)
    assert not (~close & ~region).any()


def test_streaming_statistics_agree_with_the_two_pass_mode(golden):
    """EnsembleStats(streaming=True): Welford accumulators instead of the stacked fields -- O(1) memory in the number of
    pairs -- give the reference's two-pass moments to rounding on the fields of the reference's own run (g9), and the
    whole statistics table (gradients included) to 1e-9 of each column's scale; the default mode stays bit-identical
    (the tests above)."""
    from torchpiv_amd.runner import EnsembleStats
    g = golden("g9_stats")
    n = int(np.asarray(g["n_pairs"]).reshape(-1)[0]) if "n_pairs" in g.files else None
    us = [g[k] for k in sorted(k for k in g.files if k.startswith("pair") and k.endswith("_u"))]
    vs = [g[k] for k in sorted(k for k in g.files if k.startswith("pair") and k.endswith("_v"))]
    assert us and len(us) == len(vs) and (n is None or n == len(us))
    two, one = EnsembleStats(), EnsembleStats(streaming=True)
    for u, v in zip(us, vs):
        two.add(u, v)
        one.add(u, v)
    assert one.n == two.n
    scale = max(float(np.abs(np.stack(us)).max()), float(np.abs(np.stack(vs)).max()))
    for a, b in zip(one.moments(), two.moments()):
        assert np.abs(a - b).max() <= 1e-12 * max(scale, scale * scale)
    x, y = g["pair0_x"], g["pair0_y"]
    ta, tb = one.table(x, y), two.table(x, y)
    assert list(ta) == list(tb)
    for k in ta:
        assert np.abs(ta[k] - tb[k]).max() <= 1e-9 * max(1.0, float(np.abs(tb[k]).max())), k
