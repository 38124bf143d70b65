"""Thin tensor-level wrappers over the C ABI: PyTorch-ROCm tensors in, tensors out.

PyTorch is used for device memory and streams only; all arithmetic of the hot path
happens in libtorchpiv_hip.so.  Every function requires CUDA(HIP) tensors and raises
otherwise -- there is no CPU fallback.
"""
from __future__ import annotations

import ctypes as C

import numpy as np
import torch

from . import _lib
from ._lib import ITER_MODES, MODES, PRECISIONS, check, lib


def _stream() -> int:
    return torch.cuda.current_stream().cuda_stream


def _need_cuda(*tensors):
    for t in tensors:
        if t is not None and not t.is_cuda:
            raise RuntimeError("torchpiv_amd: the HIP path needs tensors on a ROCm device "
                               "(there is no CPU fallback)")


def _frames(a: torch.Tensor, b: torch.Tensor):
    _need_cuda(a, b)
    if a.dtype != torch.uint8 or b.dtype != torch.uint8:
        raise TypeError("frames must be uint8")
    if a.shape != b.shape:
        raise ValueError("frame shapes differ")
    if a.dim() == 2:
        a, b = a[None], b[None]
    if a.dim() != 3:
        raise ValueError("frames must be [H, W] or [batch, H, W]")
    return a.contiguous(), b.contiguous()


def _work(H, W, ws, ov, batch, device):
    """Caller-provided work buffer of the function-level entry points (torch's caching allocator is
    stream-aware, so concurrent calls on different streams get different buffers)."""
    n = int(lib.tpiv_work_bytes(H, W, ws, ov, batch))
    buf = torch.empty(max(n, 16), dtype=torch.uint8, device=device)
    return buf, n


def _precision(precision):
    try:
        return PRECISIONS[precision]
    except KeyError:
        raise KeyError(f"precision must be one of {sorted(PRECISIONS)}, got {precision!r}") from None


def field_shape(H, W, ws, ov):
    """get_field_shape, PIVbackend.py:425-456."""
    nr, nc = C.c_int(), C.c_int()
    check(lib.tpiv_field_shape(H, W, ws, ov, C.byref(nr), C.byref(nc)))
    return nr.value, nc.value


def coordinates_1d(H, W, ws, ov):
    """Axis vectors of get_coordinates, PIVbackend.py:522-597."""
    nr, nc = field_shape(H, W, ws, ov)
    x = np.empty(nc, dtype=np.float64)
    y = np.empty(nr, dtype=np.float64)
    check(lib.tpiv_coordinates(H, W, ws, ov, x.ctypes.data_as(C.POINTER(C.c_double)),
                               y.ctypes.data_as(C.POINTER(C.c_double))))
    return x, y


def spline_matrix(xc: np.ndarray, xf: np.ndarray) -> np.ndarray:
    """1-D operator of the RectBivariateSpline predictor (host, float64)."""
    xc = np.ascontiguousarray(xc, dtype=np.float64)
    xf = np.ascontiguousarray(xf, dtype=np.float64)
    A = np.empty((xf.size, xc.size), dtype=np.float64)
    P = C.POINTER(C.c_double)
    check(lib.tpiv_spline_matrix(xc.size, xc.ctypes.data_as(P), xf.size, xf.ctypes.data_as(P),
                                 A.ctypes.data_as(P)))
    return A


def pass1(a, b, ws, ov, val_ratio=1.2, val_win=3, precision="exact"):
    """Device part of extended_search_area_piv(validate=True). Returns u, v (float64) and
    invalid (uint8), each [batch, n_rows, n_cols], on the frames' device.
    precision: "exact" (default, like OfflinePIV: exact integer correlation sums at the cells that reach the result, 64x64
    windows -- csrc/xcorr_exact.hip; other sizes run as "f64"), "f64" (float64 transforms as in PIVbackend.py:513-514), "reference"
    (the same kernel here) or "fast" (float32 transforms, about 1e-6 px from the reference; opt-in)."""
    a, b = _frames(a, b)
    B, H, W = a.shape
    prec = _precision(precision)
    nr, nc = field_shape(H, W, ws, ov)
    u = torch.empty(B, nr, nc, dtype=torch.float64, device=a.device)
    v = torch.empty_like(u)
    inv = torch.empty(B, nr, nc, dtype=torch.uint8, device=a.device)
    with torch.cuda.device(a.device):
        work, nbytes = _work(H, W, ws, ov, B, a.device)
        check(lib.tpiv_pass1(a.data_ptr(), b.data_ptr(), B, H, W, ws, ov, val_ratio, val_win, prec,
                             u.data_ptr(), v.data_ptr(), inv.data_ptr(), work.data_ptr(), nbytes, _stream()))
    return u, v, inv


def predict(mode, Ay, Ax, u_c, v_c, inv_c):
    """Spline predictor of one iteration: returns u0, v0 (zeroed where invalid), u2, v2."""
    _need_cuda(Ay, Ax, u_c, v_c, inv_c)
    B, nrc, ncc = u_c.shape
    nrf, ncf = Ay.shape[0], Ax.shape[0]
    dev = u_c.device
    work = torch.empty(B * 3 * nrc * ncf, dtype=torch.float64, device=dev)
    outs = [torch.empty(B, nrf, ncf, dtype=torch.float64, device=dev) for _ in range(4)]
    with torch.cuda.device(dev):
        check(lib.tpiv_predict(MODES[mode], B, nrc, ncc, nrf, ncf, Ay.data_ptr(), Ax.data_ptr(),
                               u_c.contiguous().data_ptr(), v_c.contiguous().data_ptr(),
                               inv_c.contiguous().data_ptr(), work.data_ptr(),
                               *[o.data_ptr() for o in outs], _stream()))
    return outs


def iterate(mode, a, b, ws, ov, u0, v0, u2, v2, val_ratio=1.2, val_win=3, want_raw=False, precision="exact"):
    """Device part of piv_iteration_{DWS,CWS}.__call__ after the predictor.  Shifted passes run in float32 like the
    reference's (B:249-257) at every precision; precision="reference" additionally keeps the reference's operation
    order in the CWS bilinear sampling (bit-identical staged windows), "exact" (default), "f64" and "fast" use the lerp form."""
    prec = _precision(precision)
    a, b = _frames(a, b)
    if mode == "CWS_Fast" and u2 is None:        # B:599-675: the shift is u0 / 2 inside the window; no u2 field
        u2, v2 = u0, v0
    _need_cuda(u0, v0, u2, v2)
    B, H, W = a.shape
    nr, nc = field_shape(H, W, ws, ov)
    dev = a.device
    u = torch.empty(B, nr, nc, dtype=torch.float64, device=dev)
    v = torch.empty_like(u)
    inv = torch.empty(B, nr, nc, dtype=torch.uint8, device=dev)
    du = torch.empty_like(u) if want_raw else None
    dv = torch.empty_like(u) if want_raw else None
    with torch.cuda.device(dev):
        work, nbytes = _work(H, W, ws, ov, B, dev)
        check(lib.tpiv_iter(ITER_MODES[mode], a.data_ptr(), b.data_ptr(), B, H, W, ws, ov,
                            u0.contiguous().data_ptr(), v0.contiguous().data_ptr(),
                            u2.contiguous().data_ptr(), v2.contiguous().data_ptr(),
                            val_ratio, val_win, prec, u.data_ptr(), v.data_ptr(), inv.data_ptr(),
                            du.data_ptr() if want_raw else None, dv.data_ptr() if want_raw else None,
                            work.data_ptr(), nbytes, _stream()))
    if want_raw:
        return u, v, inv, du, dv
    return u, v, inv


def debug_pass(mode, a, b, ws, ov, u2=None, v2=None, precision="reference"):
    """Test hook: one pass plus the staged windows and the correlation maps (shifted passes at
    `precision`: "reference" = the reference's operation order, bit-identical windows)."""
    prec = _precision(precision)
    a, b = _frames(a, b)
    B, H, W = a.shape
    nr, nc = field_shape(H, W, ws, ov)
    dev = a.device
    N = nr * nc
    u = torch.empty(B, nr, nc, dtype=torch.float64, device=dev)
    v = torch.empty_like(u)
    inv = torch.empty(B, nr, nc, dtype=torch.uint8, device=dev)
    win = torch.empty(B, N, 2, ws, ws, dtype=torch.float32, device=dev)
    corr = torch.empty(B, N, ws, ws, dtype=torch.float32, device=dev)
    m = 0 if mode in (0, None, "PASS1") else MODES[mode]
    zero = torch.zeros(B, nr, nc, dtype=torch.float64, device=dev) if m else None
    with torch.cuda.device(dev):
        work, nbytes = _work(H, W, ws, ov, B, dev)
        check(lib.tpiv_debug_pass(m, prec, a.data_ptr(), b.data_ptr(), B, H, W, ws, ov,
                                  u2.contiguous().data_ptr() if u2 is not None else None,
                                  v2.contiguous().data_ptr() if v2 is not None else None,
                                  zero.data_ptr() if zero is not None else None,
                                  u.data_ptr(), v.data_ptr(), inv.data_ptr(), win.data_ptr(),
                                  corr.data_ptr(), work.data_ptr(), nbytes, _stream()))
    return u, v, inv, win, corr


def debug_peaks(maps: torch.Tensor, val_ratio=1.2, val_win=3, planar=False):
    """Test hook: the kernels' peak analysis on hand-made correlation maps [n, ws, ws], ws in
    8/16/32/64/128.  planar selects the LDS layout of the three-wavefront tile kernels."""
    _need_cuda(maps)
    maps = maps.contiguous().float()
    n, ws = maps.shape[0], maps.shape[-1]
    u = torch.empty(n, dtype=torch.float64, device=maps.device)
    v = torch.empty_like(u)
    inv = torch.empty(n, dtype=torch.uint8, device=maps.device)
    work = torch.empty(max(n * 32, 16), dtype=torch.uint8, device=maps.device)
    with torch.cuda.device(maps.device):
        check(lib.tpiv_debug_peaks(maps.data_ptr(), n, ws, int(planar), float(val_ratio), int(val_win),
                                   u.data_ptr(), v.data_ptr(), inv.data_ptr(), work.data_ptr(), n * 32, _stream()))
    return u, v, inv


def postval(u, v, inv):
    """Device part of the post-validation (PIVbackend.py:884-892) for a batch, IN PLACE on u, v
    (float64 [B, nr, nc]): border interpolation, ring / hole census, fills that do not depend on the
    Delaunay triangulation.  Returns (cls uint8 [B, nr, nc], counts int32 [B, 4] = holes, ring,
    ambiguous, general); see include/torchpiv_hip.h tpiv_postval."""
    _need_cuda(u, v, inv)
    if u.dtype != torch.float64 or v.dtype != torch.float64 or inv.dtype != torch.uint8:
        raise TypeError("postval: u, v float64 and invalid uint8")
    if not (u.is_contiguous() and v.is_contiguous() and inv.is_contiguous()) or u.dim() != 3 \
            or u.shape != v.shape or u.shape != inv.shape:
        raise ValueError("postval: contiguous [batch, n_rows, n_cols] tensors of one shape")
    B, nr, nc = u.shape
    cls = torch.empty(B, nr, nc, dtype=torch.uint8, device=u.device)
    counts = torch.empty(B, 4, dtype=torch.int32, device=u.device)
    with torch.cuda.device(u.device):
        check(lib.tpiv_postval(u.data_ptr(), v.data_ptr(), inv.data_ptr(), B, nr, nc, cls.data_ptr(),
                               counts.data_ptr(), _stream()))
    return cls, counts


def postval_compact(u, v, cls, counts):
    """After postval: the ring cells (with their u, v) and the hole cells of the pairs that need the host
    triangulation, packed pair after pair in np.argwhere order (tpiv_postval_compact).  Returns (offsets int32
    [2, B + 1], ring_rc int32 [R, 2], ring_uv float64 [R, 2], hole_rc int32 [Hc, 2]) with the capacities R = B *
    ceil(cells / 4), Hc = B * cells; the used prefixes are offsets[0, B] and offsets[1, B] entries long."""
    _need_cuda(u, v, cls, counts)
    B, nr, nc = u.shape
    cells = nr * nc
    offsets = torch.empty(2, B + 1, dtype=torch.int32, device=u.device)
    ring_rc = torch.empty(B * ((cells + 3) // 4), 2, dtype=torch.int32, device=u.device)
    ring_uv = torch.empty(B * ((cells + 3) // 4), 2, dtype=torch.float64, device=u.device)
    hole_rc = torch.empty(B * cells, 2, dtype=torch.int32, device=u.device)
    with torch.cuda.device(u.device):
        check(lib.tpiv_postval_compact(u.data_ptr(), v.data_ptr(), cls.data_ptr(), counts.data_ptr(), B, nr, nc,
                                       offsets.data_ptr(), ring_rc.data_ptr(), ring_uv.data_ptr(), hole_rc.data_ptr(),
                                       _stream()))
    return offsets, ring_rc, ring_uv, hole_rc


def finish_fields(u, v, scale, dt):
    """(flip(u, rows) * scale / dt * 1000, -flip(v, rows) * scale / dt * 1000) for a batch of float64 fields on the GPU,
    with the reference's expression (PIVbackend.py:894-898) evaluated left to right: bit-identical to numpy's."""
    _need_cuda(u, v)
    if u.dtype != torch.float64 or v.dtype != torch.float64 or u.shape != v.shape or u.dim() != 3 \
            or not (u.is_contiguous() and v.is_contiguous()):
        raise ValueError("finish_fields: two contiguous float64 tensors [batch, n_rows, n_cols] of one shape")
    B, nr, nc = u.shape
    fu, fv = torch.empty_like(u), torch.empty_like(v)
    with torch.cuda.device(u.device):
        check(lib.tpiv_finish_fields(u.data_ptr(), v.data_ptr(), B, nr, nc, float(scale), float(dt), fu.data_ptr(),
                                     fv.data_ptr(), _stream()))
    return fu, fv


def ensemble_moments(U, V):
    """(mean u, mean v, <u'u'>, <v'v'>, <u'v'>) of stacked fields U, V float64 [n, ...] on the GPU, accumulated in
    stack order like numpy (tpiv_ensemble_moments)."""
    _need_cuda(U, V)
    if U.dtype != torch.float64 or V.dtype != torch.float64 or U.shape != V.shape or U.dim() < 2 or U.shape[0] < 1:
        raise ValueError("ensemble_moments: two float64 stacks [n >= 1, ...] of one shape")
    U, V = U.contiguous(), V.contiguous()
    n, cells = U.shape[0], U[0].numel()
    out = torch.empty((5,) + tuple(U.shape[1:]), dtype=torch.float64, device=U.device)
    with torch.cuda.device(U.device):
        check(lib.tpiv_ensemble_moments(U.data_ptr(), V.data_ptr(), n, cells, out.data_ptr(), _stream()))
    return tuple(out.unbind(0))


def bmp_unpack(raw, desc, lut, H, W, out=None):
    """Device unpack of raw uncompressed BMP files (tpiv_bmp_unpack).  raw uint8 [bytes] on the GPU, desc
    int64 [n, 6] and lut uint8 [n, 256] (device), see include/torchpiv_hip.h.  Returns uint8 [n, H, W]."""
    _need_cuda(raw, desc, lut)
    n = desc.shape[0]
    if desc.dtype != torch.int64 or lut.dtype != torch.uint8 or raw.dtype != torch.uint8 or tuple(desc.shape) != (n, 6) \
            or tuple(lut.shape) != (n, 256) or not (desc.is_contiguous() and lut.is_contiguous() and raw.is_contiguous()):
        raise ValueError("bmp_unpack: raw uint8 [bytes], desc int64 [n, 6], lut uint8 [n, 256], all contiguous")
    if out is None:
        out = torch.empty(n, H, W, dtype=torch.uint8, device=raw.device)
    elif out.dtype != torch.uint8 or tuple(out.shape) != (n, H, W) or not out.is_contiguous() or out.device != raw.device:
        raise ValueError("bmp_unpack: out must be a contiguous uint8 [n, H, W] tensor on the same device")
    with torch.cuda.device(raw.device):
        check(lib.tpiv_bmp_unpack(raw.data_ptr(), desc.data_ptr(), lut.data_ptr(), n, H, W, out.data_ptr(), _stream()))
    return out


class Plan:
    """The multipass pipeline of OfflinePIV.__call__ (PIVbackend.py:873-882) for batches of
    pairs resident on one GPU.  Owns the device workspace; `run` only enqueues kernels."""

    def __init__(self, H, W, ws, ov, n_pass=1, mode="CWS", pass_scale=2.0, val_ratio=1.2,
                 val_win=3, max_batch=1, device=None, precision="exact"):
        if not torch.cuda.is_available():
            raise RuntimeError("torchpiv_amd.Plan needs a ROCm device (there is no CPU fallback)")
        self.device = torch.device("cuda", torch.cuda.current_device()) if device is None \
            else torch.device(device)
        if n_pass > 1 and mode not in MODES:
            raise KeyError(mode)
        self._h = C.c_void_p()
        self.H, self.W, self.max_batch = H, W, max_batch
        self.precision = precision
        prec = _precision(precision)
        if self.device.index is None:
            self.device = torch.device("cuda", torch.cuda.current_device())
        with torch.cuda.device(self.device):
            check(lib.tpiv_plan_create(C.byref(self._h), H, W, ws, ov, n_pass, MODES.get(mode, 0),
                                       float(pass_scale), float(val_ratio), int(val_win),
                                       int(max_batch), prec))
        self.n_pass = lib.tpiv_plan_n_pass(self._h)
        self.geometry = []
        for p in range(self.n_pass):
            w, o, nr, nc = C.c_int(), C.c_int(), C.c_int(), C.c_int()
            check(lib.tpiv_plan_pass_geometry(self._h, p, C.byref(w), C.byref(o), C.byref(nr),
                                              C.byref(nc)))
            self.geometry.append((w.value, o.value, nr.value, nc.value))

    def close(self):
        h = getattr(self, "_h", None)
        if h is not None and h.value and lib is not None:     # lib is None at interpreter teardown
            lib.tpiv_plan_destroy(h)
            self._h = C.c_void_p()

    __del__ = close

    @property
    def out_shape(self):
        return self.geometry[-1][2], self.geometry[-1][3]

    def run(self, a, b, out=None):
        """a, b uint8 [batch, H, W] (or [H, W]) on the plan's device.  Returns u, v float64 and
        invalid uint8, [batch, n_rows, n_cols] of the last pass (async on the current stream)."""
        a, b = _frames(a, b)
        B = a.shape[0]
        if a.shape[1] != self.H or a.shape[2] != self.W:
            raise ValueError("frame shape differs from the plan's")
        if a.device != self.device or b.device != self.device:
            raise ValueError(f"frames live on {a.device}, the plan on {self.device}")
        if B > self.max_batch:
            raise ValueError(f"batch {B} exceeds the plan's max_batch {self.max_batch}")
        nr, nc = self.out_shape
        if out is None:
            u = torch.empty(B, nr, nc, dtype=torch.float64, device=a.device)
            v = torch.empty_like(u)
            inv = torch.empty(B, nr, nc, dtype=torch.uint8, device=a.device)
        else:
            u, v, inv = out
            for t, dt, name in ((u, torch.float64, "u"), (v, torch.float64, "v"), (inv, torch.uint8, "invalid")):
                if not isinstance(t, torch.Tensor) or t.dtype != dt:
                    raise TypeError(f"out[{name}] must be a {dt} tensor")
                if t.device != self.device:
                    raise ValueError(f"out[{name}] lives on {t.device}, the plan on {self.device}")
                if not t.is_contiguous() or t.dim() != 3 or t.shape[0] < B or tuple(t.shape[1:]) != (nr, nc):
                    raise ValueError(f"out[{name}] must be contiguous [>= {B}, {nr}, {nc}], got {tuple(t.shape)}")
        with torch.cuda.device(self.device):
            check(lib.tpiv_plan_run(self._h, a.data_ptr(), b.data_ptr(), B, u.data_ptr(),
                                    v.data_ptr(), inv.data_ptr(), _stream()))
        return u, v, inv

    def kernel_name(self, p):
        """Name of the cross-correlation kernel pass p launches (bench / profile labels)."""
        buf = C.create_string_buffer(128)
        lib.tpiv_plan_kernel_name(self._h, p, buf, 128)
        return buf.value.decode()

    def exact_capable(self):
        """precision="exact" and a first-pass window size the exact scheme covers (every even size from 8 to 128)?"""
        ws = self.geometry[0][0]
        return self.precision == "exact" and 8 <= ws <= 128 and ws % 2 == 0

    def exact_fallbacks(self):
        """precision="exact": windows of the last run whose first pass took the float64 transform (waits for the device)."""
        n = C.c_longlong()
        check(lib.tpiv_plan_exact_fallbacks(self._h, C.byref(n)))
        return n.value

    def exact_timing(self):
        """precision="exact": pass1_xcorr of the last get_timing() taken apart (mean ms): the float32 locating pass, the exact
        refinement, the float64 pass of the undecided windows, finalize."""
        arr = (C.c_double * 4)()
        check(lib.tpiv_plan_exact_timing(self._h, arr))
        return dict(zip(("locate_f32", "refine_exact", "undecided_f64", "finalize"), list(arr)))

    def debug_predict(self, p, u_c, v_c, inv_c):
        """Test hook: the plan's banded predictor of pass p on given coarse fields."""
        B = u_c.shape[0]
        _, _, nr, nc = self.geometry[p]
        outs = [torch.empty(B, nr, nc, dtype=torch.float64, device=self.device) for _ in range(4)]
        with torch.cuda.device(self.device):
            check(lib.tpiv_plan_debug_predict(self._h, p, B, u_c.contiguous().data_ptr(),
                                              v_c.contiguous().data_ptr(), inv_c.contiguous().data_ptr(),
                                              *[o.data_ptr() for o in outs], _stream()))
        return outs

    def set_timing(self, enable: bool):
        """Bracket every kernel of run() with hipEvents on the launch stream (bench.py)."""
        check(lib.tpiv_plan_set_timing(self._h, 1 if enable else 0))

    def get_timing(self):
        """({slot name: mean ms}, n_runs) for the runs since the last call; waits for them."""
        n = 2 * self.n_pass - 1
        arr = (C.c_double * n)()
        runs = C.c_int()
        check(lib.tpiv_plan_get_timing(self._h, arr, n, C.byref(runs)))
        names = ["pass1_xcorr"]
        for p in range(1, self.n_pass):
            names += [f"pass{p + 1}_predict", f"pass{p + 1}_xcorr"]
        return dict(zip(names, list(arr))), runs.value

    def pass_fields(self, p, batch):
        """Fields pass p (< n_pass-1) left in the workspace by the last run (copies)."""
        pu, pv, pi = C.c_void_p(), C.c_void_p(), C.c_void_p()
        check(lib.tpiv_plan_pass_fields(self._h, p, C.byref(pu), C.byref(pv), C.byref(pi)))
        _, _, nr, nc = self.geometry[p]
        n = batch * nr * nc
        u = torch.empty(batch, nr, nc, dtype=torch.float64, device=self.device)
        v = torch.empty_like(u)
        inv = torch.empty(batch, nr, nc, dtype=torch.uint8, device=self.device)
        torch.cuda.synchronize(self.device)
        hip = C.CDLL("libamdhip64.so")
        for dst, src, nbytes in ((u, pu, n * 8), (v, pv, n * 8), (inv, pi, n)):
            rc = hip.hipMemcpy(C.c_void_p(dst.data_ptr()), src, C.c_size_t(nbytes), 3)  # D2D
            if rc != 0:
                raise _lib.HipError(f"hipMemcpy failed: {rc}")
        return u, v, inv
