"""CPU-side tests: C-ABI symbols, geometry, spline operators, decoding, pairing, post-validation,
error mapping, FFT codelets.  No GPU compute is called here."""
import ctypes as C
import os
import re
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_cabi_exports_every_declared_symbol():
    """The library must load and export every function include/torchpiv_hip.h declares."""
    hdr = open(os.path.join(ROOT, "include", "torchpiv_hip.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = set(re.findall(r"\b(tpiv_[a-z0-9_]+)\s*\(", hdr))
    assert len(declared) >= 15
    from torchpiv_amd import _lib
    for name in sorted(declared):
        assert hasattr(_lib.lib, name), name
    assert declared == set(_lib.SIGNATURES), declared ^ set(_lib.SIGNATURES)
    assert _lib.lib.tpiv_version() == _lib.ABI_VERSION == 2
    # the production library keeps no state: the stamp setter of the diagnostic build is not in it
    assert not hasattr(_lib.lib, "tpiv_debug_set_stamps")
    assert _lib.PRECISIONS == {"fast": 0, "reference": 1, "f64": 2, "exact": 3}


def test_reference_helper_names():
    """atoi / natural_keys of PlotterFunctions.py:27-37 (the reference's own behaviour on the same inputs)."""
    from torchpiv_amd import io as pio
    assert pio.atoi("123") == 123 and pio.atoi("abc") == "abc" and pio.atoi("") == ""
    assert pio.natural_keys("image10_a.bmp") == ["image", 10, "_a.bmp"]
    assert sorted(["im10", "im9", "im1"], key=pio.natural_keys) == ["im1", "im9", "im10"]


def test_geometry_matches_reference(golden):
    from torchpiv_amd import backend, engine
    g = golden("g1_geometry")
    for H, W, ws, ov in g["cases"]:
        key = f"{H}_{W}_{ws}_{ov}"
        assert tuple(g["fs_" + key]) == engine.field_shape(int(H), int(W), int(ws), int(ov))
        assert np.array_equal(backend.get_field_shape((H, W), ws, ov), g["fs_" + key])
        x, y = backend.get_coordinates((H, W), ws, ov)
        assert np.array_equal(x[0, :], g["x_" + key]) and np.array_equal(y[:, 0], g["y_" + key])
        assert x.shape == (g["fs_" + key][0], g["fs_" + key][1])


def test_error_codes_map_to_reference_exceptions():
    from torchpiv_amd import engine
    with pytest.raises(ValueError, match="Overlap has to be smaller"):
        engine.field_shape(64, 64, 32, 32)
    with pytest.raises(ValueError, match="window size cannot be larger"):
        engine.field_shape(64, 64, 128, 0)
    with pytest.raises(ValueError):
        engine.spline_matrix(np.arange(3.0), np.arange(5.0))     # scipy needs > 3 points too


@pytest.mark.parametrize("nc,st,off", [(4, 32, 16.0), (5, 16, 8.0), (15, 32, 32.0), (63, 32, 32.0),
                                       (91, 32, 44.0), (255, 16, 16.0)])
def test_spline_operator_matches_fitpack(nc, st, off):
    """The C-side predictor operator against scipy's RectBivariateSpline (what the reference calls)."""
    from scipy import interpolate
    from torchpiv_amd import engine
    xc = off + st * np.arange(nc, dtype=np.float64)
    xf = off - st / 2 + (st / 2) * np.arange(2 * nc + 1, dtype=np.float64)   # spills over both ends
    A = engine.spline_matrix(xc, xf)
    rng = np.random.default_rng(nc)
    z = rng.standard_normal((nc, nc))
    ref = interpolate.RectBivariateSpline(xc, xc, z)(xf, xf)
    assert np.abs(A @ z @ A.T - ref).max() < 1e-12 * max(1.0, np.abs(ref).max())
    # partition of unity and interpolation at the nodes
    assert np.abs(A.sum(axis=1) - 1).max() < 1e-13
    on = np.isin(xf, xc)
    assert on.sum() == nc
    assert np.abs(A[on] - np.eye(nc)[np.searchsorted(xc, xf[on])]).max() < 1e-13


def test_bmp_decode_and_pairing(tmp_path, golden):
    from PIL import Image
    from torchpiv_amd import io as pio
    rng = np.random.default_rng(3)
    gray = rng.integers(0, 256, size=(37, 53), dtype=np.uint8)       # odd width: row padding
    Image.fromarray(gray, "L").save(tmp_path / "g.bmp")
    assert np.array_equal(pio.imdecode_gray(str(tmp_path / "g.bmp")), gray)
    rgb = rng.integers(0, 256, size=(20, 31, 3), dtype=np.uint8)
    Image.fromarray(rgb, "RGB").save(tmp_path / "c.bmp")
    got = pio.imdecode_gray(str(tmp_path / "c.bmp")).astype(int)
    want = (rgb[..., 2].astype(int) * 1868 + rgb[..., 1].astype(int) * 9617 + rgb[..., 0].astype(int) * 4899
            + 8192) >> 14
    assert np.array_equal(got, want)
    Image.fromarray(gray, "L").save(tmp_path / "g.png")
    assert np.array_equal(pio.imdecode_gray(str(tmp_path / "g.png")), gray)
    (tmp_path / "bad.bmp").write_bytes(b"not an image")
    assert pio.imdecode_gray(str(tmp_path / "bad.bmp")) is None
    # natural sort + pairing against the reference's own listing (golden g5)
    g = golden("g5_generator")
    d = tmp_path / "seq"
    d.mkdir()
    for i in range(4):
        for s in "ab":
            Image.fromarray(gray, "L").save(d / f"image{8 + i}_{s}.bmp")
    ds = pio.PIVDataset(str(d), "bmp", "pairs")
    assert [[os.path.basename(p) for p in pr] for pr in ds.img_pairs] == g["pairs_pairs"].tolist()
    ds = pio.PIVDataset(str(d), "bmp", "sequential")
    assert [[os.path.basename(p) for p in pr] for pr in ds.img_pairs] == g["seq_pairs"].tolist()
    assert len(pio.PIVDataset(str(d), "bmp", "other")) == 0
    assert len(pio.PIVDataset(str(d), "tif", "pairs")) == 0
    a, b = pio.PIVDataset(str(d), "bmp", "pairs", transform=pio.ToTensor(dtype=__import__("torch").uint8))[0]
    assert a.dtype == __import__("torch").uint8 and tuple(a.shape) == gray.shape


def test_post_validation_matches_reference(golden):
    from torchpiv_amd import backend
    g = golden("g6_kats")
    gb = backend.interpolate_boarders(g["pv_in"].copy())
    assert np.allclose(gb, g["pv_borders"], rtol=0, atol=1e-12, equal_nan=True)
    filled = backend.fillMissingValues(gb.copy())
    assert np.allclose(filled, g["pv_filled"], rtol=0, atol=1e-12, equal_nan=True)
    assert backend.fillMissingValues(np.ones((5, 5))) is None          # the dropped-clean-pair quirk
    many = np.full((6, 6), np.nan)
    many[::2, ::2] = 1.0
    assert backend.fillMissingValues(many) is None                     # too many invalid
    u, v = backend.post_validate(np.ones((4, 4)), np.ones((4, 4)), None)
    assert u is not None                                               # val None: no post-processing


def test_api_surface_and_errors_without_gpu(tmp_path):
    import torch
    import torchpiv_amd as T
    assert "cpu" in T.DeviceMap.devicies
    assert set(T.IterModMap.functions) == {"DWS", "CWS"}
    with pytest.raises(KeyError):
        T.OfflinePIV(str(tmp_path), "no-such-device", "bmp", 64, 32)
    with pytest.raises(KeyError):
        T.OfflinePIV(str(tmp_path), "cpu", "bmp", 64, 32, multipass_mode="XWS")
    empty = T.OfflinePIV(str(tmp_path), "cpu", "bmp", 64, 32)          # empty folder: fine, yields nothing
    assert len(empty) == 0 and list(empty()) == []
    from PIL import Image
    Image.fromarray(np.zeros((64, 64), np.uint8), "L").save(tmp_path / "a1.bmp")
    Image.fromarray(np.zeros((64, 64), np.uint8), "L").save(tmp_path / "a2.bmp")
    with pytest.raises(RuntimeError, match="no compute path"):
        T.OfflinePIV(str(tmp_path), "cpu", "bmp", 32, 16)               # no silent CPU fallback
    w = T.moving_window_array(torch.arange(12 * 10).reshape(12, 10), 4, 2)
    assert tuple(w.shape) == (5 * 4, 4, 4) and w[1, 0, 0] == 2 and w[4, 0, 0] == 20
    if not torch.cuda.is_available():
        with pytest.raises(RuntimeError):
            T.engine.pass1(torch.zeros(64, 64, dtype=torch.uint8), torch.zeros(64, 64, dtype=torch.uint8), 32, 16)


def test_mixed_radix_codelets_on_host(tmp_path):
    """fft_mixed.hpp (window sizes n1 * n2 with factors up to 8) compiled as host C++ against numpy.fft, both directions,
    output slots by mixed_pos."""
    sizes = [4, 6, 8, 9, 10, 12, 14, 15, 16, 18, 20, 21, 24, 25, 28, 30, 35, 36, 40, 42, 48, 49, 56, 64]
    calls = " ".join(f"run<{n},1>(); run<{n},-1>();" for n in sizes)
    src = tmp_path / "m.cpp"
    src.write_text(r"""
#include "fft_mixed.hpp"
#include <cstdio>
using namespace tpiv;
template<int N, int DIR> void run() {
  static_assert(fmx::mixed_usable(N), "size");
  cf x[N];
  for (int i = 0; i < N; ++i) { x[i].x = (float)((i*37+11)%101) - 50.f; x[i].y = (float)((i*53+7)%89) - 44.f; }
  fmx::fft_mixed<N, DIR>(x);
  printf("%d %d", N, DIR);
  for (int k = 0; k < N; ++k) printf(" %.9g %.9g", x[fmx::mixed_pos(k,N)].x, x[fmx::mixed_pos(k,N)].y);
  printf("\n");
}
int main(){ """ + calls + """ }
""")
    exe = tmp_path / "m"
    subprocess.run(["g++", "-std=c++17", "-O1", "-I", os.path.join(ROOT, "torchpiv_amd", "csrc"), str(src), "-o", str(exe)],
                   check=True)
    out = subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout.strip().splitlines()
    assert len(out) == 2 * len(sizes)
    for line in out:
        f = line.split()
        n, d = int(f[0]), int(f[1])
        got = np.array(f[2:], dtype=np.float64).reshape(n, 2)
        i = np.arange(n)
        x = ((i * 37 + 11) % 101 - 50.0) + 1j * ((i * 53 + 7) % 89 - 44.0)
        ref = np.fft.fft(x) if d > 0 else np.fft.ifft(x) * n
        err = np.abs(got[:, 0] + 1j * got[:, 1] - ref).max() / np.abs(ref).max()
        assert err < 2e-6, (n, d, err)


def test_fft_codelets_on_host(tmp_path):
    """fft_inreg.hpp compiled as host C++ against numpy.fft (forward and inverse, N = 8..128)."""
    src = tmp_path / "t.cpp"
    src.write_text(r'''
#include "fft_inreg.hpp"
#include <cstdio>
using namespace tpiv;
template<int N, int DIR> void run() {
  cf x[N];
  for (int i = 0; i < N; ++i) { x[i].x = (float)((i*37+11)%101) - 50.f; x[i].y = (float)((i*53+7)%89) - 44.f; }
  fft_inreg<N, DIR>(x);
  printf("%d %d", N, DIR);
  for (int k = 0; k < N; ++k) printf(" %.9g %.9g", x[fft_pos(k,N)].x, x[fft_pos(k,N)].y);
  printf("\n");
}
template<int N, int DIR> void runr() {      // twiddles from a TwRegs table instead of literals
  cf x[N];
  for (int i = 0; i < N; ++i) { x[i].x = (float)((i*37+11)%101) - 50.f; x[i].y = (float)((i*53+7)%89) - 44.f; }
  TwRegs<N> tw; tw.init();
  fft_inreg<N, DIR>(x, tw);
  printf("%d %d", N, DIR);
  for (int k = 0; k < N; ++k) printf(" %.9g %.9g", x[fft_pos(k,N)].x, x[fft_pos(k,N)].y);
  printf("\n");
}
int main(){ run<8,1>(); run<16,1>(); run<32,1>(); run<64,1>(); run<128,1>();
            run<8,-1>(); run<16,-1>(); run<32,-1>(); run<64,-1>(); run<128,-1>();
            runr<8,1>(); runr<16,1>(); runr<32,1>(); runr<64,1>(); runr<128,1>();
            runr<8,-1>(); runr<16,-1>(); runr<32,-1>(); runr<64,-1>(); runr<128,-1>(); }
''')
    exe = tmp_path / "t"
    subprocess.run(["g++", "-std=c++17", "-O1", "-I", os.path.join(ROOT, "torchpiv_amd", "csrc"), str(src),
                    "-o", str(exe)], check=True)
    out = subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout
    for line in out.strip().splitlines():
        t = line.split()
        N, D = int(t[0]), int(t[1])
        v = np.array(t[2:], dtype=float)
        z = v[0::2] + 1j * v[1::2]
        i = np.arange(N)
        x = ((i * 37 + 11) % 101 - 50.0) + 1j * ((i * 53 + 7) % 89 - 44.0)
        ref = np.fft.fft(x) if D > 0 else np.fft.ifft(x) * N
        assert np.abs(z - ref).max() / np.abs(ref).max() < 3e-7, (N, D)


def test_c2r_codelet_on_host(tmp_path):
    """c2r_inreg (real inverse through a half-size complex transform) against numpy.fft.irfft."""
    src = tmp_path / "t.cpp"
    src.write_text(r'''
#include "fft_inreg.hpp"
#include <cstdio>
using namespace tpiv;
template<int N> void run() {
  cf Y[N / 2 + 1], h[N / 2];
  for (int i = 0; i <= N / 2; ++i) { Y[i].x = (float)((i*37+11)%101) - 50.f; Y[i].y = (float)((i*53+7)%89) - 44.f; }
  c2r_inreg<N>(Y, h);
  printf("%d", N);
  for (int n = 0; n < N; ++n) printf(" %.9g", (n & 1) ? h[fft_pos(n / 2, N / 2)].y : h[fft_pos(n / 2, N / 2)].x);
  printf("\n");
}
int main(){ run<8>(); run<16>(); run<32>(); run<64>(); }
''')
    exe = tmp_path / "t"
    subprocess.run(["g++", "-std=c++17", "-O1", "-I", os.path.join(ROOT, "torchpiv_amd", "csrc"), str(src),
                    "-o", str(exe)], check=True)
    out = subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout
    for line in out.strip().splitlines():
        t = line.split()
        N = int(t[0])
        got = np.array(t[1:], dtype=float)
        i = np.arange(N // 2 + 1)
        Y = ((i * 37 + 11) % 101 - 50.0) + 1j * ((i * 53 + 7) % 89 - 44.0)
        ref = np.fft.irfft(Y, n=N) * N
        assert np.abs(got - ref).max() / np.abs(ref).max() < 3e-7, N


def test_image_decode_follows_opencv_rules(tmp_path):
    """Non-BMP inputs: 16-bit grey is scaled (>> 8) and colour uses OpenCV's fixed-point BGR2GRAY
    weights, as cv2.imdecode(..., IMREAD_GRAYSCALE) does in the reference (PIVbackend.py:136-137)."""
    from PIL import Image
    from torchpiv_amd import io as tio
    rng = np.random.default_rng(5)
    g16 = rng.integers(0, 65536, size=(20, 24), dtype=np.uint16)
    Image.fromarray(g16).save(tmp_path / "g16.png")
    got = tio.imdecode_gray(str(tmp_path / "g16.png"))
    assert got.dtype == np.uint8 and np.array_equal(got, (g16 >> 8).astype(np.uint8))
    rgb = rng.integers(0, 256, size=(20, 24, 3), dtype=np.uint8)
    Image.fromarray(rgb, "RGB").save(tmp_path / "c.png")
    got = tio.imdecode_gray(str(tmp_path / "c.png"))
    r, g, b = (rgb[..., i].astype(np.int64) for i in range(3))
    want = ((b * 1868 + g * 9617 + r * 4899 + 8192) >> 14).astype(np.uint8)
    assert np.array_equal(got, want)
    g8 = rng.integers(0, 256, size=(20, 24), dtype=np.uint8)
    Image.fromarray(g8, "L").save(tmp_path / "g8.tif")
    assert np.array_equal(tio.imdecode_gray(str(tmp_path / "g8.tif")), g8)


@pytest.mark.parametrize("W", [64, 128])
def test_f64_split_scheme_on_the_host(tmp_path, W):
    """The float64 first pass for 64x64 / 128x128 windows (two threads per line, 32- / 64-point codelets, radix-2 steps
    folded into the transposes; torchpiv_amd/csrc/xcorr_f64_split.hpp) run thread by thread on the CPU by
    tests/host/f64_split_harness.cpp -- the very functions the kernel calls, with the LDS plane as an array and a
    barrier as the end of a loop -- against numpy: correlation maps to 1e-13 of their maximum, arg-max, records."""
    import struct
    from oracle import piv_oracle as O
    exe = str(tmp_path / "f64h")
    subprocess.run(["g++", "-O1", "-std=c++17", "-o", exe, os.path.join(ROOT, "tests", "host", "f64_split_harness.cpp")],
                   check=True)
    rng = np.random.default_rng(W)
    n = 10 if W == 64 else 6
    A = rng.integers(0, 256, size=(n, W, W)).astype(np.uint8)
    B = rng.integers(0, 256, size=(n, W, W)).astype(np.uint8)
    shifts = {i: (3 * i - 7, 5 - 2 * i) for i in range(1, 4)}
    for i, (dy, dx) in shifts.items():                      # particle-like windows with known circular shifts
        A[i] = ((rng.random((W, W)) < 0.05) * rng.integers(60, 250, size=(W, W)) + 8).astype(np.uint8)
        B[i] = np.roll(A[i], (dy, dx), axis=(0, 1))
    A[4][:] = 0                                             # zero-mean window: "dead"
    A[5][:] = 77                                            # constant windows: flat map
    B[5][:] = 13
    A[0][:, :W // 2] = 255                                  # saturated half
    inp = struct.pack("<i", n) + b"".join(A[i].tobytes() + B[i].tobytes() for i in range(n))
    out = subprocess.run([exe, str(W)], input=inp, capture_output=True, check=True).stdout
    d = np.frombuffer(out, dtype=np.float64).reshape(n, W * W + 8)
    for i in range(n):
        aa, bb = A[i:i + 1].astype(np.float64), B[i:i + 1].astype(np.float64)
        rec = d[i, W * W:]
        if i == 4:
            assert rec[7] == 1.0                            # flagged dead: finalize writes u = v = 0, valid
            continue
        assert rec[7] == 0.0
        aa, bb = aa / aa.mean(), bb / bb.mean()
        c = O.xcorr_fft(aa, bb)[0]
        ref = c - c.min() + 1e-7
        got = d[i, :W * W].reshape(W, W)
        assert np.abs(got - ref).max() <= 1e-13 * max(ref.max(), 1.0), i
        if i == 5:
            assert int(rec[6]) == 0                         # flat map: first flat index
            continue
        m = int(rec[6])
        assert m == int(ref.argmax()) and rec[0] == got.flat[m], i
        flat = ref.reshape(1, -1).copy()
        m2 = O.second_peak(flat, np.array([m]), 3, W, W)[0]
        assert abs(rec[5] - ref.flat[m2]) <= 1e-13 * ref.max(), i
        for slot, q in ((1, m + 1), (2, m - 1), (3, m + W), (4, m - W)):
            if 0 < q < W * W - 1:
                assert rec[slot] == got.flat[q], (i, slot)
        if i in shifts:                                     # the circular shift is found exactly
            assert (m // W - W // 2, m % W - W // 2) == shifts[i]


@pytest.mark.parametrize("W,wv", [(64, 3), (64, 0), (64, 1), (64, 5), (128, 3), (128, 6)])
def test_f64_split_peak_stage_on_handmade_maps(tmp_path, W, wv):
    """The float64 kernels' peak stage (third generation: raw cells, zone rows, shift on read; xcorr_f64_split.hpp P) on
    crafted maps -- a peak at every special flat index (every one-sided fix-up of B:385-392, the last-column neighbour in
    the next row), rivals at the edges, the row wraps and the two clamps of the exclusion zone of B:346-358, ties, a
    constant map -- against numpy + the oracle's second_peak (pinned to the reference by g6): arg-max, the six record
    cells bit for bit.  val_win 5 / 6 take the path that keeps the zone rows in the plane."""
    import struct
    from oracle import piv_oracle as O
    exe = str(tmp_path / "f64h")
    subprocess.run(["g++", "-O1", "-std=c++17", "-o", exe, os.path.join(ROOT, "tests", "host", "f64_split_harness.cpp")],
                   check=True)
    rng = np.random.default_rng(7 * W + wv)
    n = W * W
    specials = [0, 1, 2, W - 2, W - 1, W, W + 1, 2 * W - 1, 2 * W, n - 2 * W, n - W - 1, n - W, n - W + 1, n - 3, n - 2,
                n - 1, n // 2 + W // 2, n // 2 + W // 2 - 1, 5 * W - 1, 6 * W, 3 * W + 3, n - 3 * W - 4, (W // 2) * W,
                (W // 2) * W + W - 1, wv * W + wv, wv * W + wv + 1, n - 1 - wv * W - wv, n - 2 - wv * W - wv]
    rivals = (None, wv + 1, -(wv + 1), wv, -wv, wv * W + wv, -(wv * W + wv), wv * W + wv + 1, (wv + 1) * W, -(wv + 1) * W,
              W - wv, W - wv - 1, -(W - wv), "first", "last")
    maps = []
    for m in specials:
        for rv in rivals:
            a = rng.random((W, W)) * 5 + 1
            a.flat[m] = 100.0
            for q, val in ((m + 1, 60.0), (m - 1, 40.0), (m + W, 55.0), (m - W, 35.0)):
                if 0 <= q < n:
                    a.flat[q] = val
            if rv is not None:
                rival = 0 if rv == "first" else (n - 1 if rv == "last" else (m + rv) % n)
                if rival != m:
                    a.flat[rival] = max(a.flat[rival], 90.0 if (rival + m) % 2 else 80.0)
            maps.append(a)
    maps += [rng.random((W, W)) * 10 - 5 for _ in range(32)]          # raw cells may be negative
    tie = np.ones((W, W))
    tie[W // 4, W // 4] = tie[W // 2 + 1, W // 2 + 2] = 50.0            # exact tie: first flat index wins
    maps += [tie, np.full((W, W), 5.0)]
    maps = np.stack(maps)
    inp = struct.pack("<ii", len(maps), wv) + maps.astype(np.float64).tobytes()
    out = subprocess.run([exe, str(W), "maps"], input=inp, capture_output=True, check=True).stdout
    rec = np.frombuffer(out, dtype=np.float64).reshape(len(maps), 8)
    for i, a in enumerate(maps):
        v = (a - a.min()) * 1.0 + 1e-7
        flat = v.reshape(1, -1)
        m = int(flat.argmax())
        assert int(rec[i, 6]) == m, (i, int(rec[i, 6]), m)
        left, right, top, bot = m + 1, m - 1, m + W, m - W                  # B:385-392
        left = m if left >= n - 1 else left
        right = m if right <= 0 else right
        top = m if top >= n - 1 else top
        bot = m if bot <= 0 else bot
        for slot, q in ((0, m), (1, left), (2, right), (3, top), (4, bot)):
            assert rec[i, slot] == flat[0, q], (i, slot)
        z = flat.copy()
        m2 = int(O.second_peak(z, np.array([m]), wv, W, W)[0])
        assert rec[i, 5] == z[0, m2], (i, m, m2, rec[i, 5], z[0, m2])      # (0.0 when the whole map is excluded)
        assert rec[i, 7] == 0.0


def test_fast_hole_fill_is_the_reference_interpolator_bit_for_bit():
    """torchpiv_amd._qhull.qhull_fill (scipy.spatial.Delaunay + the barycentric arithmetic of scipy's _interpnd spelled out
    in numpy) against the reference's literal scipy.interpolate.LinearNDInterpolator call on random hole patterns --
    isolated invalid vectors (co-circular diamonds: Qhull's tie-break decides), runs, blobs, holes at the border: every
    filled value bit-identical, NaNs (targets outside the hull) and refusals (collinear rings) alike."""
    from torchpiv_amd._qhull import qhull_fill, qhull_fill_reference
    rng = np.random.default_rng(2024)
    n_vals = n_masks = n_nan = n_refused = 0
    for trial in range(700):
        nr, nc = rng.integers(6, 70, 2)
        h = rng.random((nr, nc)) < rng.choice([0.004, 0.02, 0.05, 0.12])
        if rng.random() < 0.35:
            r, c = rng.integers(0, nr - 3), rng.integers(0, nc - 3)
            h[r:r + rng.integers(1, 4), c:c + rng.integers(1, 5)] = True
        if rng.random() < 0.1:
            h[rng.integers(0, nr), :] = True               # a whole row: the ring falls apart into two lines
        if not h.any():
            continue
        d = h.copy()
        d[1:] |= h[:-1]
        d[:-1] |= h[1:]
        d[:, 1:] |= h[:, :-1]
        d[:, :-1] |= h[:, 1:]
        pts, tg = np.argwhere(d & ~h), np.argwhere(h)
        vals = rng.standard_normal((len(pts), 2)) * 5
        ref, got = qhull_fill_reference(pts, vals, tg), qhull_fill(pts, vals, tg)
        assert (ref is None) == (got is None), trial
        if ref is None:
            n_refused += 1
            continue
        assert np.array_equal(ref, got, equal_nan=True), trial
        n_masks += 1
        n_vals += ref.size
        n_nan += int(np.isnan(ref).sum())
    print(f"  {n_masks} masks, {n_vals} filled values bit-identical ({n_nan} NaN outside the hull), {n_refused} rings refused by Qhull")
    assert n_masks > 500 and n_nan > 0 and n_refused > 0


def test_diamond_short_cut_of_the_hole_fill_is_bit_identical():
    """Fields whose holes are all ISOLATED invalid vectors (the common case): qhull_fill reads Qhull's diagonal of
    every co-circular diamond off the triangulation instead of computing barycentric transforms and walking simplices
    (_qhull._diamond_fill).  Against the reference's literal LinearNDInterpolator call and against the general path: every
    value bit-identical -- holes next to each other diagonally, at the field border (then the short cut must step aside), values
    over nine decades -- and the short cut really is the path taken."""
    from torchpiv_amd import _qhull
    rng = np.random.default_rng(77)
    taken = general = n_vals = 0
    for trial in range(300):
        nr, nc = (int(t) for t in rng.integers(8, 130, 2))
        hole = np.zeros((nr, nc), bool)
        border = rng.random() < 0.15                      # holes ON the border have fewer than four neighbours: general path
        for _ in range(int(rng.integers(1, 80))):
            lo = 0 if border else 1
            r, c = int(rng.integers(lo, nr - lo)), int(rng.integers(lo, nc - lo))
            nb = [(r + dr, c + dc) for dr, dc in ((1, 0), (-1, 0), (0, 1), (0, -1)) if 0 <= r + dr < nr and 0 <= c + dc < nc]
            if hole[r, c] or any(hole[q] for q in nb):
                continue
            hole[r, c] = True
        d = hole.copy()
        d[1:] |= hole[:-1]
        d[:-1] |= hole[1:]
        d[:, 1:] |= hole[:, :-1]
        d[:, :-1] |= hole[:, 1:]
        pts, tg = np.argwhere(d & ~hole), np.argwhere(hole)
        vals = rng.standard_normal((len(pts), 2)) * float(rng.choice([1e-6, 1e-2, 1.0, 1e3]))
        ref = _qhull.qhull_fill_reference(pts, vals, tg)
        got = _qhull.qhull_fill(pts, vals, tg)
        slow = _qhull.qhull_fill(pts, vals, tg, diamonds=False)
        assert (ref is None) == (got is None) == (slow is None), trial
        if ref is None:
            continue
        assert np.array_equal(ref, got, equal_nan=True) and np.array_equal(ref, slow, equal_nan=True), trial
        from scipy.spatial import Delaunay
        short = _qhull._diamond_fill(Delaunay(pts), pts, vals, tg)
        taken += short is not None
        general += short is None
        n_vals += ref.size
        on_border = bool(hole[0].any() or hole[-1].any() or hole[:, 0].any() or hole[:, -1].any())
        assert (short is None) == on_border, (trial, on_border)
    print(f"  {taken} fields through the short cut, {general} with a border hole through the general path, {n_vals} values bit-identical")
    assert taken > 200 and general > 10


def test_fill_workers_keep_the_order_of_the_jobs(monkeypatch):
    """The fill-worker processes of the generator (torchpiv_amd._qhull.FillWorkers): tickets collected in any order, light
    batches left in flight while a heavy one (exchanged on the spot, job by job) passes, answers equal to the in-process fill."""
    from torchpiv_amd._qhull import FillWorkers, qhull_fill
    rng = np.random.default_rng(0)

    def job(nh):
        hole = np.zeros((12, 12), bool)
        idx = rng.choice(100, nh, replace=False)
        hole[1 + idx // 10, 1 + idx % 10] = True
        pts = np.argwhere(~hole)
        return pts, rng.normal(size=(pts.shape[0], 2)), np.argwhere(hole)

    monkeypatch.setattr(FillWorkers, "LIGHT", 16000)
    fw = FillWorkers(3)
    try:
        light_a, light_b, heavy = [job(3) for _ in range(10)], [job(5) for _ in range(7)], [job(40) for _ in range(5)] * 3
        batches = [light_a, light_b, heavy, light_a[:2], []]
        tickets = [fw.submit(b) for b in batches]
        assert fw._order and tickets[2] in fw._ready                 # two kinds of ticket really were in play
        for k in (3, 0, 4, 2, 1):
            got = fw.collect(tickets[k])
            assert len(got) == len(batches[k])
            for g_, j in zip(got, batches[k]):
                assert np.array_equal(g_, qhull_fill(*j), equal_nan=True)
        assert not fw._order and not fw._ready and not fw._parts
    finally:
        fw.terminate()
    assert all(not p.is_alive() for p in fw.procs) and not fw.conns


def test_batch_reader_agrees_with_the_per_file_path(tmp_path):
    """tpiv_read_files + the vectorised header sweep (io.stage_batch) against io.stage_raw file by file: same bytes
    in the slot, same layout and grey table for the plain BMPs; everything else is handed back (None)."""
    from PIL import Image

    from torchpiv_amd import io as pio
    rng = np.random.default_rng(5)
    H, W = 37, 50
    paths = []
    for i in range(5):
        Image.fromarray(rng.integers(0, 256, size=(H, W)).astype(np.uint8), "L").save(tmp_path / f"f{i}.bmp")
        paths.append(str(tmp_path / f"f{i}.bmp"))
    pal = Image.fromarray(rng.integers(0, 256, size=(H, W)).astype(np.uint8), "P")
    pal.putpalette(list(rng.integers(0, 256, size=768)))
    pal.save(tmp_path / "p.bmp")
    Image.fromarray(rng.integers(0, 256, size=(H, W, 3)).astype(np.uint8), "RGB").save(tmp_path / "c.bmp")
    Image.fromarray(rng.integers(0, 256, size=(H + 1, W)).astype(np.uint8), "L").save(tmp_path / "other_shape.bmp")
    (tmp_path / "bad.bmp").write_bytes(b"BMnot")
    Image.fromarray(rng.integers(0, 256, size=(H, W)).astype(np.uint8), "L").save(tmp_path / "x.png")
    paths += [str(tmp_path / n) for n in ("p.bmp", "c.bmp", "other_shape.bmp", "bad.bmp", "x.png", "missing.bmp")]
    cap = 8192
    raw, raw2 = np.zeros((len(paths), cap), np.uint8), np.zeros((len(paths), cap), np.uint8)
    lays = pio.stage_batch(paths, raw, H, W, threads=3)
    assert [lay is not None for lay in lays] == [True] * 7 + [False] * 4
    for j in range(7):
        ref = pio.stage_raw(paths[j], raw2[j], H, W)
        assert lays[j][:4] == ref[:4] and np.array_equal(lays[j][4], ref[4])
        assert np.array_equal(raw[j], raw2[j])
    # a slot smaller than the file: handed back, nothing written past the slot
    small = np.zeros((1, 1024), np.uint8)
    assert pio.stage_batch(paths[:1], small, H, W) == [None]


def test_read_ahead_ring(tmp_path):
    """tpiv_reader_*: batches arrive in order and complete, never more than len(bufs) ahead; a short last batch; unreadable
    files are -1; closing mid-run stops the threads."""
    from torchpiv_amd import io as pio
    rng = np.random.default_rng(9)
    blobs, paths = [], []
    for i in range(23):
        blob = rng.integers(0, 256, size=int(rng.integers(100, 3000)), dtype=np.uint8).tobytes()
        if i == 7:
            paths.append(str(tmp_path / "nope.bin"))
            blobs.append(None)
            continue
        if i == 11:
            blob = bytes(5000)                      # larger than a slot
        (tmp_path / f"f{i}.bin").write_bytes(blob)
        paths.append(str(tmp_path / f"f{i}.bin"))
        blobs.append(blob)
    cap, per = 4096, 4
    bufs = [np.zeros((per, cap), np.uint8) for _ in range(3)]
    rd = pio.ReadAhead(paths, per, [b.ctypes.data for b in bufs], cap, threads=5)
    seen = 0
    held = 0
    while True:
        got = rd.next()
        if got is None:
            break
        k, sizes = got
        assert k == (seen // per) % 3 and len(sizes) == min(per, 23 - seen)
        for j, sz in enumerate(sizes):
            blob = blobs[seen + j]
            if blob is None or len(blob) > cap:
                assert sz == -1
            else:
                assert sz == len(blob) and bufs[k][j, :sz].tobytes() == blob
        seen += len(sizes)
        held += 1
        if held == 2:                               # hold two buffers at a time, hand the oldest back
            rd.release()
            held -= 1
    assert seen == 23
    rd.close()
    rd.close()
    # every buffer held: next() must refuse instead of waiting for ever
    rd = pio.ReadAhead(paths, per, [b.ctypes.data for b in bufs[:1]], cap, threads=2)
    assert rd.next() is not None
    with pytest.raises(ValueError):
        rd.next()
    rd.close()                                      # mid-run: threads waiting for a buffer are released
    # read=False: nothing is read, every file is handed to the per-file path
    rd = pio.ReadAhead(paths, per, [b.ctypes.data for b in bufs], cap, read=False)
    k, sizes = rd.next()
    assert k == 0 and list(sizes) == [-1] * per


def test_qhull_diamond_choice_is_not_a_local_rule():
    """Why the isolated invalid vector stays a (counted) host triangulation: its four ring points are co-circular, both
    diagonals are Delaunay, and the one SciPy/Qhull takes -- hence the value, (N + S) / 2 or (E + W) / 2 -- changes with
    a SECOND hole 20-40 cells away, i.e. it is not a function of any neighbourhood of the hole (PIVbackend.py:284-308;
    tools/research/qhull_diamonds.py).  The product therefore hands such pairs to the same Qhull (torchpiv_amd/_qhull.py)."""
    from torchpiv_amd._qhull import qhull_fill
    nr = nc = 63
    rng = np.random.default_rng(3)
    field = rng.standard_normal((nr, nc))
    seen = {}
    for far in ((5, 5), (5, 23), (11, 29), (5, 41), (47, 47), (41, 11)):
        hole = np.zeros((nr, nc), bool)
        hole[20, 20] = hole[far] = True
        ring = np.zeros_like(hole)
        ring[1:] |= hole[:-1]
        ring[:-1] |= hole[1:]
        ring[:, 1:] |= hole[:, :-1]
        ring[:, :-1] |= hole[:, 1:]
        ring &= ~hole
        val = qhull_fill(np.argwhere(ring), field[ring][:, None], np.array([[20, 20]]))[0, 0]
        ns, ew = (field[19, 20] + field[21, 20]) / 2, (field[20, 19] + field[20, 21]) / 2
        assert min(abs(val - ns), abs(val - ew)) < 1e-15
        seen["NS" if abs(val - ns) < abs(val - ew) else "EW"] = far
    assert set(seen) == {"NS", "EW"}, seen


def test_hand_issued_lds_reads_are_not_touched_before_their_wait():
    """The float64 kernels issue ds_read_b64 / ds_read_b128 through inline asm and wait with a separate counted
    s_waitcnt (xcorr_f64_split.hpp): the compiler must not read, copy or spill a destination register in between -- it
    cannot know the value has not landed.  tools/check_lds_inflight.py walks the device assembly of the
    build; first on two crafted snippets (the checker must see a violation and pass a clean sequence), then on the real
    translation unit."""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import check_lds_inflight as C
    bad = """
_ZkernelA:
	;;#ASMSTART
	ds_read_b128 v[4:7], v1 offset:16
	;;#ASMEND
	scratch_store_dwordx2 off, v[6:7], off offset:8
	;;#ASMSTART
	s_waitcnt lgkmcnt(0)
	;;#ASMEND
	s_endpgm
"""
    good = """
_ZkernelB:
	;;#ASMSTART
	ds_read_b128 v[4:7], v1 offset:16
	;;#ASMEND
	;;#ASMSTART
	ds_read_b64 v[8:9], v1 offset:32
	;;#ASMEND
	v_add_f64 v[20:21], v[10:11], v[12:13]
	;;#ASMSTART
	s_waitcnt lgkmcnt(1)
	;;#ASMEND
	v_add_f64 v[20:21], v[4:5], v[6:7]
	;;#ASMSTART
	s_waitcnt lgkmcnt(0)
	;;#ASMEND
	v_add_f64 v[20:21], v[8:9], v[6:7]
	s_endpgm
"""
    assert len(C.check(bad)[0]) == 1 and C.check(good) == ([], 2)
    assert C.main("xcorr_f64") == 0
    assert C.main("xcorr_exact") == 0          # the refinement kernel's single ds_read_b32 spans
