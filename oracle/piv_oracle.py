"""CPU oracle for the PIV cross-correlation hot path -- TEST INFRASTRUCTURE ONLY.

This file is a from-scratch restatement (numpy, plus torch-CPU for the FFT so
that the transform runs in the same precision as the reference: float64 in
pass 1, float32 in passes >= 2) of the algorithm of NikNazarov/TorchPIV's
`PIVbackend.py`.  Every function cites the reference lines it follows
(`B:` = /root/reference/src/torchPIV/PIVbackend.py).

Only `tests/`, `__graft_entry__.smoke()` and the `cpu_baseline` leg of
`bench.py` may import it, and only as the checker.  The product
(`torchpiv_amd/`) never imports it and has no CPU fallback.

Parity pin: the reference ships no tests/golden vectors for this path
(SURVEY.md section 4), so the oracle is pinned against outputs of the
reference itself, produced in the build container by
`tests/golden/make_golden.py` and committed under `tests/golden/*.npz`
(`tests/test_oracle_golden.py` checks them on every CPU test run).
"""
from __future__ import annotations

import numpy as np
import torch
from numpy.lib.stride_tricks import as_strided

EPS = 1e-7  # B:380


# --------------------------------------------------------------------------
# geometry
# --------------------------------------------------------------------------
def field_shape(image_size, ws, ov):
    """B:425-456 -- (size - ws)//(ws - ov) + 1 per axis."""
    return (np.array(image_size) - ws) // (ws - ov) + 1


def coordinates(image_size, ws, ov):
    """B:522-597 -- window-centre coordinates, grid centred by an integer shift."""
    fs = field_shape(image_size, ws, ov)
    x = np.arange(fs[-1], dtype=np.int32) * (ws - ov) + ws / 2.0
    y = np.arange(fs[-2], dtype=np.int32) * (ws - ov) + ws / 2.0
    x += (image_size[-1] - 1 - ((fs[-1] - 1) * (ws - ov) + (ws - 1))) // 2
    y += (image_size[-2] - 1 - ((fs[-2] - 1) * (ws - ov) + (ws - 1))) // 2
    return np.meshgrid(x, y)


def windows(arr: np.ndarray, ws, ov) -> np.ndarray:
    """B:220-247 -- overlapping windows anchored at pixel (0, 0), row-major
    over (grid row, grid col); returns a materialised [N, ws, ws] copy."""
    H, W = arr.shape
    st = ws - ov
    n_r = int((H - ws) / st) + 1
    n_c = int((W - ws) / st) + 1
    s0, s1 = arr.strides
    v = as_strided(arr, shape=(n_r, n_c, ws, ws), strides=(s0 * st, s1 * st, s0, s1))
    return v.reshape(-1, ws, ws)


# --------------------------------------------------------------------------
# correlation and peak analysis
# --------------------------------------------------------------------------
def xcorr_fft(aa: np.ndarray, bb: np.ndarray) -> np.ndarray:
    """B:249-257 -- fftshift(irfft2(conj(rfft2(a)) * rfft2(b))).
    uint8 input is promoted to float32 by torch.fft (as in the reference's DWS
    passes); float32 stays float32; float64 stays float64."""
    ta, tb = torch.from_numpy(np.ascontiguousarray(aa)), torch.from_numpy(np.ascontiguousarray(bb))
    c = torch.fft.fftshift(torch.fft.irfft2(torch.fft.rfft2(ta).conj() * torch.fft.rfft2(tb)),
                           dim=(-2, -1))
    return c.numpy()


def second_peak(cor: np.ndarray, m: np.ndarray, wind: int, k: int, d: int) -> np.ndarray:
    """B:346-358 -- zero (in place) the 7x7 flat-index neighbourhood of the
    first peak (flat index arithmetic, clamped to [0, k*d-1], wraps over row
    ends), then argmax again.  `cor` is [c, k*d] and is modified."""
    rows = np.arange(cor.shape[0])
    for i in range(-wind, wind + 1):
        for j in range(-wind, wind + 1):
            ids = np.clip(m + i + k * j, 0, k * d - 1)
            cor[rows, ids] = 0.0
    return _argmax_nanfirst(cor)


def _argmax_nanfirst(a2d: np.ndarray) -> np.ndarray:
    """torch.argmax semantics: first maximal index; a NaN counts as the
    maximum (first NaN wins).  numpy's argmax has the same NaN behaviour."""
    return np.argmax(a2d, axis=-1)


def corr_to_disp(corr: np.ndarray, n_rows, n_cols, validate=True, val_ratio=1.2,
                 validation_window=3):
    """B:360-422 -- peak, 3-point log-Gaussian fit by FLAT index neighbours,
    peak-to-second-peak validation.  `corr` [c, d, k] float32/float64."""
    c, d, k = corr.shape
    corr = corr + corr.dtype.type(EPS)            # B:381 (in corr's own dtype)
    flat = corr.reshape(c, -1)
    cor = flat.astype(np.float64)                 # B:382
    m = _argmax_nanfirst(flat)                    # B:383
    kd = k * d
    left, right, top, bot = m + 1, m - 1, m + k, m - k       # B:385-388 (names as in the reference)
    left = np.where(left >= kd - 1, m, left)      # B:389
    right = np.where(right <= 0, m, right)        # B:390
    top = np.where(top >= kd - 1, m, top)         # B:391
    bot = np.where(bot <= 0, m, bot)              # B:392
    rows = np.arange(c)
    cm, cl, cr, ct, cb = (cor[rows, m], cor[rows, left], cor[rows, right],
                          cor[rows, top], cor[rows, bot])
    with np.errstate(all="ignore"):
        lm, ll, lr, lt, lb = np.log(cm), np.log(cl), np.log(cr), np.log(ct), np.log(cb)
        nom1 = lr - ll                             # B:399
        den1 = 2 * (ll + lr) - 4 * lm              # B:400
        nom2 = lb - lt                             # B:401
        den2 = 2 * (lb + lt) - 4 * lm              # B:402
        v = (m // d) + nom2 / den2                 # B:404-406
        u = (m % k) + nom1 / den1                  # B:407
    mask = None
    if validate:
        work = cor if corr.dtype == np.float64 else flat.copy()
        # B:410 -- the reference zeroes `corr` in place; for float64 input `cor`
        # aliases it, but cm was gathered before and m2 lies outside the zeroed
        # set, so the aliasing is unobservable.
        m2 = second_peak(work.copy(), m, validation_window, k, d)
        with np.errstate(all="ignore"):
            mask = (cm / cor[rows, m2]) < val_ratio          # B:411
        dead = (left >= kd - 1) & (right <= 0) & (top >= kd - 1) & (bot <= 0)   # B:412
        mask = np.where(dead, True, mask).reshape(n_rows, n_cols)
    v = v - int(d / 2)                             # B:415-417
    u = u - int(k / 2)
    v = np.nan_to_num(v)                           # B:418-419
    u = np.nan_to_num(u)
    return u.reshape(n_rows, n_cols), v.reshape(n_rows, n_cols), mask


def pass1(frame_a: np.ndarray, frame_b: np.ndarray, ws=32, ov=0, validate=False,
          validation_ratio=1.2):
    """B:459-520 -- first pass: windows, divide by the window mean (float64),
    FFT cross-correlation, minus minimum, peak analysis."""
    if ov >= ws:
        raise ValueError("Overlap has to be smaller than the window_size")
    if ws > frame_a.shape[-2] or ws > frame_a.shape[-1]:
        raise ValueError("window size cannot be larger than the image")
    n_rows, n_cols = field_shape(frame_a.shape, ws, ov)
    x, y = coordinates(frame_a.shape, ws, ov)
    aa = windows(frame_a, ws, ov)
    bb = windows(frame_b, ws, ov)
    with np.errstate(all="ignore"):
        aa = aa / aa.mean(axis=(-2, -1), dtype=np.float64, keepdims=True)    # B:513
        bb = bb / bb.mean(axis=(-2, -1), dtype=np.float64, keepdims=True)    # B:514
    corr = xcorr_fft(aa, bb)
    with np.errstate(all="ignore"):
        corr = corr - corr.min(axis=(-2, -1), keepdims=True)                  # B:518
    u, v, mask = corr_to_disp(corr, n_rows, n_cols, validate, validation_ratio)
    return u, v, x, y, mask


# --------------------------------------------------------------------------
# window shifting (passes >= 2)
# --------------------------------------------------------------------------
def window_index(frame_shape, ws, ov) -> np.ndarray:
    """B:684-687 / B:751-754 -- flat pixel index of every window element."""
    H, W = frame_shape
    return windows(np.arange(H * W, dtype=np.int64).reshape(H, W), ws, ov)


def shift_dws(arr: np.ndarray, grid: np.ndarray, vel_x: np.ndarray, vel_y: np.ndarray) -> np.ndarray:
    """B:197-216 -- integer window shift on the FLAT index, clamped flat."""
    W = arr.shape[-1]
    g = grid + vel_y * W + vel_x
    np.clip(g, 0, arr.size - 1, out=g)
    return arr.reshape(-1)[g]


def shift_cws(arr: np.ndarray, grid: np.ndarray, vel_x: np.ndarray, vel_y: np.ndarray) -> np.ndarray:
    """B:147-194 -- bilinear window shift in float32 with flat-index clamping
    and the 'either coordinate integral => nearest sample' quirk (B:170,193).
    arr uint8 [H, W]; grid int64 [c, ws, ws]; vel float32 [c, 1, 1]."""
    W = arr.shape[-1]
    f32 = np.float32
    gy, gx = grid // W, grid % W
    ny = gy.astype(f32) + vel_y.astype(f32)        # int64 + float32 tensor -> float32
    nx = gx.astype(f32) + vel_x.astype(f32)
    ux = np.ceil(nx).astype(np.int64)
    uy = np.ceil(ny).astype(np.int64)
    dx = np.floor(nx).astype(np.int64)
    dy = np.floor(ny).astype(np.int64)
    mask = (ux - dx) * (uy - dy) == 0
    n = arr.size
    q12 = np.clip(uy * W + dx, 0, n - 1)
    q11 = np.clip(dy * W + dx, 0, n - 1)
    q22 = np.clip(uy * W + ux, 0, n - 1)
    q21 = np.clip(dy * W + ux, 0, n - 1)
    flat = arr.reshape(-1)
    f11, f12, f21, f22 = (flat[q11].astype(f32), flat[q12].astype(f32),
                          flat[q21].astype(f32), flat[q22].astype(f32))
    uxf, uyf, dxf, dyf = ux.astype(f32), uy.astype(f32), dx.astype(f32), dy.astype(f32)
    out = (f11 * (uxf - nx) * (uyf - ny)
           + f21 * (nx - dxf) * (uyf - ny)
           + f12 * (uxf - nx) * (ny - dyf)
           + f22 * (nx - dxf) * (ny - dyf))       # B:187-192, left-to-right, float32
    out = out.astype(f32)
    out[mask] = f11[mask]
    return out


# --------------------------------------------------------------------------
# predictor (third-party FITPACK via scipy, as the reference calls it)
# --------------------------------------------------------------------------
def spline_predict(y0, x0, z, slice_y, slice_x):
    """B:700-704 -- scipy.interpolate.RectBivariateSpline(y, x, z)(yf, xf):
    bicubic interpolating spline, evaluation points clamped to the data range
    by FITPACK.  scipy is unpinned in the reference (setup.cfg:29); 1.15.3 here."""
    from scipy import interpolate
    return interpolate.RectBivariateSpline(y0[:, 0], x0[0, :], z)(slice_y, slice_x)


class IterPass:
    """Common part of piv_iteration_DWS / piv_iteration_CWS (B:677-740, B:744-812)."""

    def __init__(self, frame_shape, ws, ov):
        self.ws, self.ov = ws, ov
        self.n_rows, self.n_cols = field_shape(frame_shape, ws, ov)
        self.x, self.y = coordinates(frame_shape, ws, ov)
        self.slice_x, self.slice_y = self.x[0, :], self.y[:, 0]
        self.idx = window_index(frame_shape, ws, ov)

    def _predict(self, x0, y0, u0, v0, validation_mask):
        u0 = spline_predict(y0, x0, u0, self.slice_y, self.slice_x)
        v0 = spline_predict(y0, x0, v0, self.slice_y, self.slice_x)
        return u0, v0

    def _finish(self, aa, bb, u0, v0, u2, v2, validate):
        corr = xcorr_fft(aa, bb)
        corr = corr - corr.min(axis=(-2, -1), keepdims=True)
        du, dv, val = corr_to_disp(corr, self.n_rows, self.n_cols, validate)
        v = 2 * v2 + dv
        u = 2 * u2 + du
        mask_u = (du > u0) * (np.rint(u0) > 0)
        mask_v = (dv > v0) * (np.rint(v0) > 0)
        if val is not None:
            mask_u[val] = True
            mask_v[val] = True
        v[mask_v] = v0[mask_v]
        u[mask_u] = u0[mask_u]
        return u, v, self.x, self.y, val, du, dv


class IterCWS(IterPass):
    """B:677-740."""

    def __call__(self, frame_a, frame_b, x0, y0, u0, v0, validation_mask, debug=False):
        u0, v0 = self._predict(x0, y0, u0, v0, validation_mask)
        u2 = u0 / 2            # B:705-706: BEFORE the invalid-zeroing
        v2 = v0 / 2
        validate = False
        if validation_mask is not None:
            validate = True
            val = spline_predict(y0, x0, validation_mask, self.slice_y, self.slice_x) >= .5
            u0[val] = 0.0
            v0[val] = 0.0
        u2t = u2.astype(np.float32).reshape(-1)[:, None, None]
        v2t = v2.astype(np.float32).reshape(-1)[:, None, None]
        aa = shift_cws(frame_a, self.idx, -u2t, -v2t)
        bb = shift_cws(frame_b, self.idx, u2t, v2t)
        out = self._finish(aa, bb, u0, v0, u2, v2, validate)
        if debug:
            return out + (u0, v0, u2, v2)
        return out[:5]


class IterDWS(IterPass):
    """B:744-812."""

    def __call__(self, frame_a, frame_b, x0, y0, u0, v0, validation_mask, debug=False):
        u0, v0 = self._predict(x0, y0, u0, v0, validation_mask)
        validate = False
        if validation_mask is not None:
            validate = True
            val = spline_predict(y0, x0, validation_mask, self.slice_y, self.slice_x) >= .5
            u0[val] = 0.0
            v0[val] = 0.0
        v2 = np.rint(v0 / 2)   # B:782-785: AFTER the zeroing, round-half-even
        u2 = np.rint(u0 / 2)
        u2t = u2.astype(np.int64).reshape(-1)[:, None, None]
        v2t = v2.astype(np.int64).reshape(-1)[:, None, None]
        aa = shift_dws(frame_a, self.idx, -u2t, -v2t)
        bb = shift_dws(frame_b, self.idx, u2t, v2t)
        out = self._finish(aa, bb, u0, v0, np.rint(u2), np.rint(v2), validate)
        if debug:
            return out + (u0, v0, u2, v2)
        return out[:5]


class IterCWSFast(IterPass):
    """B:599-675 piv_iteration_CWS_Fast (not registered in the reference's IterModMap, so OfflinePIV
    never reaches it): every window is resampled INSIDE ITSELF by -/+ u0/2 with torch's bicubic
    grid_sample (border padding, align_corners=False -- third-party arithmetic, called as the reference
    calls it), normalised by its float32 mean, correlated, and the result is u = u0 + du."""

    def __call__(self, frame_a, frame_b, x0, y0, u0, v0, validation_mask, debug=False):
        import torch.nn.functional as F
        u0, v0 = self._predict(x0, y0, u0, v0, validation_mask)
        validate = False
        if validation_mask is not None:
            validate = True
            val = spline_predict(y0, x0, validation_mask, self.slice_y, self.slice_x) >= .5
            u0[val] = 0.0                                  # B:630-633
            v0[val] = 0.0
        ws = self.ws
        aa = torch.from_numpy(windows(frame_a, ws, self.ov).copy())[:, None].float()
        bb = torch.from_numpy(windows(frame_b, ws, self.ov).copy())[:, None].float()
        theta = torch.tensor([[1., 0., 0.], [0., 1., 0.]]).repeat(aa.shape[0], 1, 1)
        uflat, vflat = torch.from_numpy(u0.flatten()), torch.from_numpy(v0.flatten())
        theta[:, 1, 2] = -vflat / ws                        # B:644-648
        theta[:, 0, 2] = -uflat / ws
        aa = F.grid_sample(aa, F.affine_grid(theta, aa.size(), align_corners=False), mode="bicubic",
                           padding_mode="border", align_corners=False)
        theta[:, 1, 2] = vflat / ws                         # B:650-653
        theta[:, 0, 2] = uflat / ws
        bb = F.grid_sample(bb, F.affine_grid(theta, bb.size(), align_corners=False), mode="bicubic",
                           padding_mode="border", align_corners=False)
        aa = aa / torch.mean(aa, (-2, -1), dtype=torch.float32, keepdim=True)      # B:656-657
        bb = bb / torch.mean(bb, (-2, -1), dtype=torch.float32, keepdim=True)
        corr = xcorr_fft(aa.numpy()[:, 0], bb.numpy()[:, 0])
        with np.errstate(all="ignore"):
            corr = corr - corr.min(axis=(-2, -1), keepdims=True)
        du, dv, val = corr_to_disp(corr, self.n_rows, self.n_cols, validate)
        v = v0 + dv                                         # B:663-664
        u = u0 + du
        mask_u = (du > u0) * (np.rint(u0) > 0)
        mask_v = (dv > v0) * (np.rint(v0) > 0)
        if val is not None:
            mask_u[val] = True
            mask_v[val] = True
        v[mask_v] = v0[mask_v]
        u[mask_u] = u0[mask_u]
        if debug:
            return u, v, self.x, self.y, val, du, dv, u0, v0, aa.numpy()[:, 0], bb.numpy()[:, 0]
        return u, v, self.x, self.y, val


ITER = {"DWS": IterDWS, "CWS": IterCWS}      # B:814-818 (IterCWSFast is unreachable there, as in the reference)


# --------------------------------------------------------------------------
# post-validation (host side of the generator, B:884-892)
# --------------------------------------------------------------------------
def interp_borders(vec: np.ndarray) -> np.ndarray:
    """B:328-344 -- 1-D linear interpolation of NaNs on the four borders."""
    if not np.isnan(vec).any():
        return vec
    for sl in ((0, slice(None)), (-1, slice(None)), (slice(None), 0), (slice(None), -1)):
        line = vec[sl]
        nans = np.isnan(line)
        if not nans.all():
            idx = np.arange(line.size)
            line[nans] = np.interp(idx[nans], idx[~nans], line[~nans])
            vec[sl] = line
    return vec


def _dilate_cross(mask: np.ndarray) -> np.ndarray:
    """cv2.dilate with the 3x3 MORPH_ELLIPSE element (= 4-connected cross),
    constant zero border (B:275-279)."""
    out = mask.copy()
    out[1:, :] |= mask[:-1, :]
    out[:-1, :] |= mask[1:, :]
    out[:, 1:] |= mask[:, :-1]
    out[:, :-1] |= mask[:, 1:]
    return out


def fill_missing(t: np.ndarray):
    """B:266-308 -- Delaunay linear fill of interior holes from the ring of
    valid neighbours; None on failure (including the 'no invalid vectors =>
    no points => exception' case, B:303-304) or when too many are invalid."""
    from scipy import interpolate
    invalid = np.isnan(t)
    ring = _dilate_cross(invalid) & ~invalid
    points = np.argwhere(ring)
    values = t[ring]
    if points.size < ring.size / 2:
        try:
            interp = interpolate.LinearNDInterpolator(points, values)
            t[invalid] = interp(np.argwhere(invalid))
        except Exception:
            return None
    else:
        return None
    return t


def offline_piv(pairs, ws, ov, multipass=1, mode="CWS", dt=1, scale=1.0, multipass_scale=2.0):
    """B:862-903 -- the generator body for an iterable of (frame_a, frame_b)
    uint8 arrays; yields (x, y, u, v) float64 arrays; dropped pairs yield nothing."""
    iters = None
    for a, b in pairs:
        if a is None or b is None:
            continue
        if iters is None:
            iters, w, o = [], ws, ov
            for _ in range(multipass - 1):
                w = int(w // multipass_scale)
                o = int(o // multipass_scale)
                iters.append(ITER[mode](a.shape, w, o))
        u, v, x, y, val = pass1(a, b, ws, ov, validate=True)
        for it in iters:
            u, v, x, y, val = it(a, b, x, y, u, v, val)
        if val is not None:
            u[val] = np.nan
            v[val] = np.nan
            u = interp_borders(u)
            v = interp_borders(v)
            u = fill_missing(u)
            v = fill_missing(v)
            if u is None or v is None:
                continue
        u = np.flip(u, axis=0)
        v = -np.flip(v, axis=0)
        u = u * scale / dt * 1000
        v = v * scale / dt * 1000
        yield x * scale, y * scale, u, v
