"""The headless runner (ensemble statistics + export formats of the reference's GUI worker)."""
import os

import numpy as np
import pytest


def _reference_table(x, y, u_inst, v_inst):
    """The statistics exactly as workers.py:85-118 computes them from the stacked fields."""
    u_inst, v_inst = np.stack(u_inst), np.stack(v_inst)
    avg_u = np.mean(u_inst, axis=0, dtype=np.float64)
    avg_v = np.mean(v_inst, axis=0, dtype=np.float64)
    uu = np.mean((u_inst - avg_u) ** 2, axis=0, dtype=np.float64)
    vv = np.mean((v_inst - avg_v) ** 2, axis=0, dtype=np.float64)
    uv = np.mean((u_inst - avg_u) * (v_inst - avg_v), axis=0, dtype=np.float64)
    mid_i, mid_j = x.shape[-2] // 2, x.shape[-1] // 2
    dx = (x[mid_i, mid_j + 1] - x[mid_i, mid_j]) / 1000
    dy = (y[mid_i + 1, mid_j] - y[mid_i, mid_j]) / 1000
    dUy, dUx = np.gradient(avg_u, dx, dy, edge_order=2)
    dVy, dVx = np.gradient(avg_v, dx, dy, edge_order=2)
    return {"Vx[m/s]": avg_u, "Vy[m/s]": avg_v, "(vx-Vx)(vy-Vy)[m^2/s^2]": uv, "(vx-Vx)^2[m^2/s^2]": uu,
            "(vy-Vy)^2[m^2/s^2]": vv, "dVx/dx[1/s]": dUx, "dVx/dy[1/s]": dUy, "dVy/dx[1/s]": dVx,
            "dVy/dy[1/s]": dVy, "W[1/s]": dVx - dUy, "S[1/s]": dVx + dUy}


def test_streaming_statistics_equal_two_pass():
    from torchpiv_amd.runner import EnsembleStats
    rng = np.random.default_rng(5)
    x, y = np.meshgrid(16.0 + 8 * np.arange(11), 12.0 + 8 * np.arange(9))
    us = [rng.standard_normal((9, 11)) * 3 + 10 for _ in range(7)]
    vs = [rng.standard_normal((9, 11)) * 2 - 4 for _ in range(7)]
    st = EnsembleStats()
    for u, v in zip(us, vs):
        st.add(u, v)
    t = st.table(x, y)
    ref = _reference_table(x, y, us, vs)
    assert list(t)[:4] == ["x[mm]", "y[mm]", "Vx[m/s]", "Vy[m/s]"] and len(t) == 13
    for k, v in ref.items():
        assert np.allclose(t[k], v, rtol=1e-10, atol=1e-10), k


def test_export_formats(tmp_path):
    from torchpiv_amd.runner import save_binary, save_table, uniquify
    d = str(tmp_path / "Out")
    data = {"x[mm]": np.arange(6.0).reshape(2, 3), "Vx[m/s]": np.arange(6.0).reshape(2, 3) / 7}
    p1 = save_table("run_pair.txt", d, data.copy())
    p2 = save_table("run_pair.txt", d, data.copy())          # never overwrites: " (1)" is appended
    assert os.path.basename(p1) == "run_pair.txt" and os.path.basename(p2) == "run_pair (1).txt"
    lines = open(p1).read().splitlines()
    assert lines[0] == "x[mm], Vx[m/s]" and lines[2] == "1.000000, 0.142857" and len(lines) == 7
    pb = save_binary("run_pair.npy", d, data.copy())
    arr = np.load(pb)
    assert arr.shape == (2, 2, 3) and np.array_equal(arr[0], data["x[mm]"])
    assert uniquify(str(tmp_path / "nothing.txt")).endswith("nothing.txt")


@pytest.mark.gpu
def test_run_folder_matches_generator(tmp_path, golden):
    from PIL import Image
    import torchpiv_amd as T
    from torchpiv_amd.runner import run_folder
    g = golden("g5_generator")
    d = tmp_path / "pairs"
    d.mkdir()
    for i, (a, b) in enumerate(zip(g["frames_a"], g["frames_b"])):
        Image.fromarray(a, "L").save(d / f"image{8 + i}_a.bmp")
        Image.fromarray(b, "L").save(d / f"image{8 + i}_b.bmp")
    kw = dict(wind_size=32, overlap=16, multipass=3, multipass_mode="CWS", dt=2, scale=0.5)
    out = str(tmp_path / "Out")
    table, n = run_folder(str(d), "cuda:0", "bmp", save_opt="Save all text", save_dir=out, batch_size=3, **kw)
    res = list(T.OfflinePIV(str(d), "cuda:0", "bmp", **kw)())
    assert n == len(res) == 4
    ref = _reference_table(res[0][0], res[0][1], [r[2] for r in res], [r[3] for r in res])
    for k, v in ref.items():
        assert np.allclose(table[k], v, rtol=1e-9, atol=1e-9, equal_nan=True), k
    files = sorted(os.listdir(out))
    assert "pairs_statistics.txt" in files and sum(f.startswith("pairs_pair") for f in files) == 4
    hdr = open(os.path.join(out, "pairs_statistics.txt")).readline().strip().split(", ")
    assert hdr == list(table.keys())
