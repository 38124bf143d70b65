"""Drop-in host side of the engine: the reference's public names with the reference's
signatures, defaults and error behaviour (NikNazarov/TorchPIV, src/torchPIV/PIVbackend.py,
cited as B:), driving the HIP library through torchpiv_amd.engine.

What runs where
  * device (libtorchpiv_hip.so): every pass of a pair -- windows, shift, FFT correlation,
    peak, validation, predictor, combine -- with the fields staying on the GPU between passes
    (the reference does three D2H copies and one H2D per pass);
  * device, batched (tpiv_postval): the post-validation of B:884-892 as far as it does not depend on
    Qhull's tie-breaking -- NaN-out, border interpolation, the ring / hole census (both drop decisions
    follow from it, so dropped pairs never leave the GPU) and every hole whose Delaunay-linear value is
    fixed by its N/S or E/W neighbours;
  * host (this file): dataset/decoding, the Delaunay hole fill (scipy/Qhull, as in the reference) for
    the pairs that still hold an ambiguous or wide hole -- counted in OfflinePIV.stats --, flip and
    unit scaling, the generator protocol.

There is no CPU compute path: device="cpu" raises.
"""
from __future__ import annotations

from time import time
from typing import Generator

import numpy as np
import torch

from . import engine
from ._lib import PRECISIONS
from ._qhull import qhull_fill, qhull_fill_many      # numpy/scipy-only module: the fill worker processes import nothing else
from .io import PIVDataset, ToTensor, natural_keys  # noqa: F401  (re-exported like the reference)


# ----------------------------------------------------------------------------------------
# device / mode maps (B:13-18, B:814-818)
# ----------------------------------------------------------------------------------------
class _DeviceDict(dict):
    """name -> torch.device.  The reference keys CUDA devices by torch.cuda.get_device_name(i),
    so eight identical MI355X collapse to one key (the last index wins).  That behaviour is
    kept for those names; 'cuda', 'cuda:N' and plain integers are accepted as well."""

    def __init__(self):
        super().__init__()
        self._filled = False

    def _fill(self):
        if not self._filled:
            self._filled = True
            for i in range(torch.cuda.device_count()):
                try:
                    dict.__setitem__(self, torch.cuda.get_device_name(i), torch.device("cuda", i))
                except Exception:       # no usable GPU in this process
                    break
            dict.__setitem__(self, "cpu", torch.device("cpu"))

    def __getitem__(self, key):
        self._fill()
        if dict.__contains__(self, key):
            return dict.__getitem__(self, key)
        if isinstance(key, torch.device):
            return key
        if isinstance(key, int) and 0 <= key < torch.cuda.device_count():
            return torch.device("cuda", key)
        if isinstance(key, str) and (key == "cuda" or key.startswith("cuda:")):
            d = torch.device(key)
            return torch.device("cuda", d.index if d.index is not None else torch.cuda.current_device())
        raise KeyError(key)

    def keys(self):
        self._fill()
        return dict.keys(self)

    def __iter__(self):
        self._fill()
        return dict.__iter__(self)

    def __len__(self):
        self._fill()
        return dict.__len__(self)

    def __contains__(self, key):
        try:
            self[key]
            return True
        except KeyError:
            return False


class DeviceMap:
    devicies = _DeviceDict()          # (sic) the reference's attribute name


def _require_gpu(device: torch.device) -> torch.device:
    if device.type != "cuda":
        raise RuntimeError("torchpiv_amd runs on MI355X only: device='cpu' has no compute path here "
                           "(use the upstream TorchPIV for CPU runs)")
    return device


# ----------------------------------------------------------------------------------------
# geometry (B:425-456, B:522-597, B:220-247)
# ----------------------------------------------------------------------------------------
def get_field_shape(image_size, search_area_size, overlap):
    return (np.array(image_size) - search_area_size) // (search_area_size - overlap) + 1


def get_coordinates(image_size, search_area_size, overlap):
    """(x, y) meshgrid of window-centre coordinates, as the reference returns it."""
    x, y = engine.coordinates_1d(int(image_size[-2]), int(image_size[-1]), int(search_area_size),
                                 int(overlap))
    return np.meshgrid(x, y)


def moving_window_array(array: torch.Tensor, window_size, overlap) -> torch.Tensor:
    """Overlapping windows [N, ws, ws] of a 2-D tensor (a strided view made contiguous).  The
    kernels never materialise this; it is kept for callers that used the reference's helper."""
    H, W = array.shape[-2], array.shape[-1]
    st = window_size - overlap
    n_r, n_c = (H - window_size) // st + 1, (W - window_size) // st + 1
    return torch.as_strided(array, size=(n_r, n_c, window_size, window_size),
                            stride=(W * st, st, W, 1)).reshape(-1, window_size, window_size)


# ----------------------------------------------------------------------------------------
# pass 1 (B:459-520)
# ----------------------------------------------------------------------------------------
def extended_search_area_piv(frame_a, frame_b, window_size=32, overlap=0, validate: bool = False,
                             validation_ratio: float = 1.2, precision: str = "exact"):
    """First-pass PIV of one pair.  frame_a / frame_b: uint8 tensors [H, W] on a ROCm device.
    Returns (u, v, x, y, mask) as numpy arrays like the reference (mask None if not validate).
    Raises ValueError for overlap >= window_size or a window larger than the image.
    precision (extension): "exact" (default) = the map cells behind the result as exact integer correlation sums (64x64
    windows; within 1e-14 px of the reference's float64 pass, bit-identical for most windows), "f64" / "reference" = float64
    transforms like B:513-514 (what "exact" runs for other window sizes), "fast" = float32 transforms."""
    H, W = frame_a.shape[-2], frame_a.shape[-1]
    u, v, inv = engine.pass1(frame_a, frame_b, int(window_size), int(overlap),
                             val_ratio=float(validation_ratio), precision=precision)
    x, y = get_coordinates((H, W), window_size, overlap)
    mask = inv[0].cpu().numpy().astype(bool) if validate else None
    return u[0].cpu().numpy(), v[0].cpu().numpy(), x, y, mask


# ----------------------------------------------------------------------------------------
# multipass iterations (B:677-740, B:744-812)
# ----------------------------------------------------------------------------------------
class _piv_iteration:
    mode = None

    def __init__(self, frame_shape, wind_size, overlap, device) -> None:
        self.frame_shape = (int(frame_shape[-2]), int(frame_shape[-1]))
        self.wind_size, self.overlap = int(wind_size), int(overlap)
        self.n_rows, self.n_cols = get_field_shape(self.frame_shape, self.wind_size, self.overlap)
        self.device = _require_gpu(DeviceMap.devicies[device] if not isinstance(device, torch.device)
                                   else device)
        self.x, self.y = get_coordinates(self.frame_shape, self.wind_size, self.overlap)
        self.slice_x, self.slice_y = self.x[0, :], self.y[:, 0]
        self._ops = {}

    def _operators(self, y0, x0):
        key = (y0[:, 0].tobytes(), x0[0, :].tobytes())
        if key not in self._ops:
            Ay = torch.from_numpy(engine.spline_matrix(y0[:, 0], self.slice_y)).to(self.device)
            Ax = torch.from_numpy(engine.spline_matrix(x0[0, :], self.slice_x)).to(self.device)
            self._ops = {key: (Ay, Ax)}
        return self._ops[key]

    def __call__(self, frame_a, frame_b, x0, y0, u0, v0, validation_mask):
        Ay, Ax = self._operators(np.asarray(y0, dtype=np.float64), np.asarray(x0, dtype=np.float64))
        dev = self.device
        validate = validation_mask is not None
        inv_c = (torch.from_numpy(np.ascontiguousarray(validation_mask).astype(np.uint8)) if validate
                 else torch.zeros(u0.shape, dtype=torch.uint8))
        u_c = torch.from_numpy(np.ascontiguousarray(u0, dtype=np.float64)).to(dev)[None]
        v_c = torch.from_numpy(np.ascontiguousarray(v0, dtype=np.float64)).to(dev)[None]
        # (CWS_Fast: the predictor fields after the invalid-zeroing are all it needs, B:626-633)
        p_u0, p_v0, p_u2, p_v2 = engine.predict("CWS" if self.mode == "CWS_Fast" else self.mode, Ay, Ax, u_c, v_c,
                                                inv_c.to(dev)[None])
        # validate=False: the reference skips the peak-ratio test (val stays None)
        ratio = 1.2 if validate else float("-inf")
        u, v, inv = engine.iterate(self.mode, frame_a.to(dev), frame_b.to(dev), self.wind_size,
                                   self.overlap, p_u0, p_v0, p_u2, p_v2, val_ratio=ratio)
        val = inv[0].cpu().numpy().astype(bool) if validate else None
        return u[0].cpu().numpy(), v[0].cpu().numpy(), self.x, self.y, val


class piv_iteration_CWS(_piv_iteration):
    """Continuous (bilinear) window shift iteration, B:677-740."""
    mode = "CWS"


class piv_iteration_DWS(_piv_iteration):
    """Discrete (integer) window shift iteration, B:744-812."""
    mode = "DWS"


class piv_iteration_CWS_Fast(_piv_iteration):
    """The reference's bicubic window-deformation iteration (B:599-675): every window is resampled inside
    itself by -/+ u0/2 (torch's grid_sample, mode "bicubic", border padding), normalised by its mean and
    correlated; u = u0 + du.  As in the reference it is NOT in IterModMap (OfflinePIV never uses it) and its
    call takes three more arguments, which the reference ignores in favour of the constructor's except for
    the window geometry."""
    mode = "CWS_Fast"

    def __call__(self, frame_a, frame_b, x0, y0, u0, v0, validation_mask, wind_size=None, overlap=None, device=None):
        if wind_size is not None and (int(wind_size), int(overlap if overlap is not None else self.overlap)) != \
                (self.wind_size, self.overlap):
            raise ValueError("piv_iteration_CWS_Fast: wind_size / overlap differ from the constructor's")
        return super().__call__(frame_a, frame_b, x0, y0, u0, v0, validation_mask)


class IterModMap:
    functions = {"DWS": piv_iteration_DWS, "CWS": piv_iteration_CWS}


# ----------------------------------------------------------------------------------------
# post-validation on the host (B:266-344, B:884-892)
# ----------------------------------------------------------------------------------------
def nan_helper(y):
    """(NaN mask, index helper) of a 1-D array -- the reference's helper of the same name (B:311-326)."""
    mask = np.isnan(y)

    def indices(selector):
        return np.flatnonzero(selector)
    return mask, indices


def interpolate_boarders(vec: np.ndarray) -> np.ndarray:
    """Linear 1-D interpolation of NaNs along the four borders (an all-NaN border is left)."""
    if not np.isnan(vec).any():
        return vec
    for line in (vec[0, :], vec[-1, :], vec[:, 0], vec[:, -1]):      # views: edits land in vec
        nans = np.isnan(line)
        if not nans.all():
            idx = np.arange(line.size)
            line[nans] = np.interp(idx[nans], idx[~nans], line[~nans])
    return vec


def getPixelsForInterp(img):
    """(ring, invalid): ring = valid cells 4-adjacent to an invalid (NaN) cell.  The reference
    dilates with OpenCV's 3x3 MORPH_ELLIPSE element, which is the 4-connected cross, with a
    constant zero border (B:275-279)."""
    invalid = np.isnan(img)
    dil = invalid.copy()
    dil[1:, :] |= invalid[:-1, :]
    dil[:-1, :] |= invalid[1:, :]
    dil[:, 1:] |= invalid[:, :-1]
    dil[:, :-1] |= invalid[:, 1:]
    return dil & ~invalid, invalid


TOO_MANY_MSG = "Warning! to many false vectors"      # the reference's stdout line (B:306), spelling included


def fillMissingValues(target_for_interp, interpolator=None):
    """Fill NaN holes by Delaunay-linear interpolation from the ring of valid neighbours (B:284-308), in place.
    None -- the pair is to be dropped -- when the ring holds a quarter of the cells or more (the reference's
    size test on the flattened point list, with its warning line), when there is nothing to interpolate from
    (no invalid vector at all: the reference's interpolator raises on zero points and its bare `except`
    swallows that, B:300-304) or when Qhull refuses the ring.  `interpolator`: optional stand-in for scipy's
    LinearNDInterpolator class."""
    ring, holes = getPixelsForInterp(target_for_interp)
    n_ring = int(np.count_nonzero(ring))
    if 4 * n_ring >= ring.size:                       # 2 coordinates per ring point against size / 2
        print(TOO_MANY_MSG)
        return None
    where_ring, where_holes = np.argwhere(ring), np.argwhere(holes)
    if interpolator is None:
        vals = qhull_fill(where_ring, target_for_interp[ring], where_holes)
    else:
        try:
            vals = interpolator(where_ring, target_for_interp[ring])(where_holes)
        except Exception:
            vals = None
    if vals is None:
        return None
    target_for_interp[holes] = vals
    return target_for_interp


def post_validate(u, v, val):
    """B:884-892: NaN-out invalid vectors, interpolate borders, fill holes.  (None, None) when
    the pair has to be dropped."""
    if val is not None:
        u[val] = np.nan
        v[val] = np.nan
        u = interpolate_boarders(u)
        v = interpolate_boarders(v)
        u = fillMissingValues(u)
        v = fillMissingValues(v)
        if u is None or v is None:
            return None, None
    return u, v


def _ring_of(hole: np.ndarray) -> np.ndarray:
    """Valid cells 4-adjacent to a hole (getPixelsForInterp, B:266-282: 3x3 cross, zero border)."""
    ring = np.zeros_like(hole)
    ring[1:, :] |= hole[:-1, :]
    ring[:-1, :] |= hole[1:, :]
    ring[:, 1:] |= hole[:, :-1]
    ring[:, :-1] |= hole[:, 1:]
    ring &= ~hole
    return ring


def fill_holes_host(u: np.ndarray, v: np.ndarray, hole: np.ndarray, solve=qhull_fill):
    """fillMissingValues (B:284-308) for u AND v with ONE triangulation: both fields carry the same
    holes, so the reference's two interpolators triangulate the same ring points in the same order
    and apply the same barycentric weights; evaluating a two-column interpolator is bit-identical.
    u, v are filled in place; returns False when the reference would drop the pair (Qhull refuses
    the ring, e.g. collinear points; the size test is done by the caller from the ring count)."""
    ring = _ring_of(hole)
    vals = solve(np.argwhere(ring), np.stack([u[ring], v[ring]], axis=1), np.argwhere(hole))
    if vals is None:
        return False
    u[hole] = vals[:, 0]
    v[hole] = vals[:, 1]
    return True


def free_cuda_memory():
    torch.cuda.empty_cache()


# ----------------------------------------------------------------------------------------
# the generator API (B:824-903)
# ----------------------------------------------------------------------------------------
class OfflinePIV:
    """for x, y, u, v in OfflinePIV(folder, device, file_fmt, wind_size, overlap, ...)(): ...

    Same constructor signature, defaults, __len__ and generator protocol as the reference:
    yields float64 numpy arrays [n_rows, n_cols] of the last pass; u is flipped along axis 0
    and v flipped and negated; u, v are scaled by scale/dt*1000 and x, y by scale; pairs that
    cannot be decoded or whose hole fill fails are skipped silently.
    """

    verbose = False          # the reference prints timing lines to stdout; off by default here

    def __init__(self, folder: str, device: str, file_fmt: str, wind_size: int, overlap: int,
                 multipass: int = 1, multipass_mode: str = "CWS", dt: int = 1, scale: float = 1.,
                 multipass_scale: float = 2., folder_mode: str = "pairs", precision: str = "exact",
                 validation_ratio: float = 1.2, validation_window: int = 3) -> None:
        # precision (extension, keyword after the reference's arguments).  "exact" (default): as "f64", with the map cells
        # that reach the result of a 64x64 first pass evaluated as exact integer correlation sums instead of through a
        # float64 FFT (csrc/xcorr_exact.hip: within 1e-14 px of the reference's float64 pass 1, about 1.5x the rate of
        # "f64"; other first-pass sizes run as "f64").  "f64": the reference's own arithmetic types in every transform --
        # pass 1 in float64 (B:513-514), later passes float32 with a float64 epilogue.
        # "reference": the same plus the reference's operation order in the CWS sampling (bit-identical staged
        # windows).  "fast": pass 1 in float32 too (~1e-6 px from the float64 pass 1, about 1.9x the rate).
        # validation_ratio / validation_window (extensions): the constants the reference hides inside
        # correlation_to_displacement (B:364-365: val_ratio=1.2, validation_window=3), same defaults.
        if precision not in PRECISIONS:
            raise KeyError(precision)
        self._precision = precision
        self._val_ratio, self._val_win = float(validation_ratio), int(validation_window)
        self._wind_size = wind_size
        self._overlap = overlap
        self._dt = dt
        self._iter = multipass
        self._iter_scale = multipass_scale
        self._scale = scale
        self._device = DeviceMap.devicies[device]                       # KeyError like B:845
        self._dataset = PIVDataset(folder, file_fmt, folder_mode, transform=ToTensor(dtype=torch.uint8))
        self._iter_function = IterModMap.functions[multipass_mode]      # KeyError like B:850
        self._mode = multipass_mode
        self._plan = None
        self._single_plans = {}          # plans of the one-pair path, per frame shape
        self.reset_stats()
        if not self:
            return
        _require_gpu(self._device)

    def __len__(self) -> int:
        return len(self._dataset)

    def frame_shape(self):
        """(H, W) of the first decodable pair, None for an empty / undecodable folder."""
        for i in range(len(self._dataset)):
            a, _ = self._dataset[i]
            if a is not None:
                return tuple(a.shape)
        return None

    def _get_plan(self, H, W, max_batch=1):
        if (self._plan is None or (self._plan.H, self._plan.W) != (H, W)
                or self._plan.max_batch < max_batch):
            if self._plan is not None:
                self._plan.close()
            self._plan = engine.Plan(H, W, int(self._wind_size), int(self._overlap),
                                     n_pass=max(1, int(self._iter)), mode=self._mode,
                                     pass_scale=self._iter_scale, max_batch=max_batch,
                                     val_ratio=self._val_ratio, val_win=self._val_win,
                                     device=self._device, precision=self._precision)
        return self._plan

    def reset_stats(self):
        """Counters of the post-validation: pairs seen, dropped for 'no invalid vector' / 'too many
        false vectors' (decided on the device), finished on the device alone, sent to the host
        triangulation (and dropped there because Qhull refused the ring)."""
        self.stats = {"pairs": 0, "dropped_no_invalid": 0, "dropped_too_many": 0, "device_complete": 0,
                      "host_fallback": 0, "dropped_by_qhull": 0,
                      # host_fallback by hole class: pairs whose only undetermined cells are co-circular
                      # diamonds (isolated invalid vectors: Qhull's tie-break decides), and pairs that hold a wider hole
                      "host_fallback_diamonds_only": 0, "host_fallback_wide_holes": 0}

    fill_workers = 0         # > 0: the host triangulations of a batch run in that many worker processes
    pipeline_depth = 2       # batched() over files: launches in flight before a batch's results are collected
    read_threads = 8         # file reader threads of batched() (a page-cache read into pinned memory runs at ~3 GB/s per thread)
    device_out = False       # batched(): yield u, v as float64 tensors ON THE DEVICE (the finished fields of B:894-898, hole
                             # fills scattered in from the host) instead of numpy arrays -- for callers that go on with them
                             # on the GPU (dist.run_sharded: the end-of-run gather over xGMI)

    def auto_host_config(self, ranks_on_node=None):
        """Size the reader threads and the fill-worker processes from the cores this rank really has (affinity mask and
        container quota divided by the ranks of the node, torchpiv_amd.hostcfg) instead of the single-GPU defaults.
        Returns the budget dict."""
        from . import hostcfg
        b = hostcfg.host_budget(ranks_on_node)
        self.read_threads, self.fill_workers = b["read_threads"], b["fill_workers"]
        return b

    def _fill_pool(self):
        """The fill-worker processes (_qhull.FillWorkers), started on first use and again when fill_workers changes."""
        if self.fill_workers <= 0:
            return None
        pool = getattr(self, "_pool", None)
        if pool is None or getattr(self, "_pool_size", 0) != self.fill_workers:
            if pool is not None:
                pool.terminate()
            # spawned: the workers never see this process's HIP state; they only run scipy on small arrays (with ONE BLAS
            # thread each: _qhull.single_thread_blas)
            from ._qhull import FillWorkers
            self._pool = pool = FillWorkers(self.fill_workers)
            self._pool_size = self.fill_workers
        return pool

    def close(self):
        pool = getattr(self, "_pool", None)
        if pool is not None:
            pool.terminate()
            self._pool = None
        if getattr(self, "_plan", None) is not None:
            self._plan.close()
            self._plan = None
        rd = getattr(self, "_reader", None)
        if rd is not None:
            rd.close()
            self._reader = None
        self._stage, self._stage_key = None, None
        self._raw_dev, self._raw_dev_key = None, None
        for pl in getattr(self, "_single_plans", {}).values():
            pl.close()
        self._single_plans = {}

    RING_CAP = 256           # ring / hole cells per pair (batch average) that ride on the first, asynchronous copy

    def _post_submit(self, u, v, inv, want_raw=False):
        """Device half of B:884-898 for a batch of final fields (u, v float64 [n, nr, nc], modified in
        place; inv uint8): tpiv_postval (NaN-out, border interpolation, census, triangulation-free fills),
        tpiv_postval_compact (the ring points with their values and the hole cells of the pairs that need Qhull, cut out
        and packed in np.argwhere order on the device), the flip and the unit scaling of B:894-898 (the reference's own
        float64 expressions -- u * scale / dt * 1000, three correctly rounded operations -- evaluated by the device on
        the whole batch: bit-identical, and the host is spared five passes over every field), then ASYNCHRONOUS copies of
        the census, the packed lists and the finished fields into pinned memory, on a stream of their own.  Nothing here
        waits for the GPU: the caller may enqueue the next batch before it collects this one.  want_raw: also the raw
        (unflipped, unscaled) fields, for callers of the function-level API."""
        cls, counts = engine.postval(u, v, inv)
        offsets, ring_rc, ring_uv, hole_rc = engine.postval_compact(u, v, cls, counts)
        fu, fv = engine.finish_fields(u, v, self._scale, self._dt)
        n = u.shape[0]
        cap_r, cap_h = min(ring_rc.shape[0], n * self.RING_CAP), min(hole_rc.shape[0], n * self.RING_CAP)
        src = {"counts": counts, "offsets": offsets, "ring_rc": ring_rc[:cap_r], "ring_uv": ring_uv[:cap_r],
               "hole_rc": hole_rc[:cap_h], "fu": fu, "fv": fv}
        if want_raw:
            src["u"], src["v"] = u, v
        host = {k: torch.empty(t.shape, dtype=t.dtype, pin_memory=True) for k, t in src.items()}
        # the copies go down on a stream of their own, behind an event of the compute stream: the next batch's passes
        # start while they run.  (The small kernels above stay on the compute stream: on the side stream too they made
        # the whole generator slower -- 9.1 -> 8.2 k pairs/s, same box.)
        cur = torch.cuda.current_stream(u.device)
        down = getattr(self, "_down_stream", None)
        if down is None or down.device != u.device:
            down = self._down_stream = torch.cuda.Stream(u.device)
        ready = torch.cuda.Event()
        ready.record(cur)
        with torch.cuda.stream(down):
            down.wait_event(ready)
            for k, t in src.items():
                host[k].copy_(t, non_blocking=True)
            done = torch.cuda.Event()
            done.record(down)
        # (every device tensor the copy stream reads stays referenced until the ticket is collected, i.e. past `done`;
        #  the full lists serve the overflow path)
        return done, host, (src, ring_rc, ring_uv, hole_rc, u, v)

    def _post_extract(self, ticket):
        """Host half, first stage: drop decisions from the census; the pairs that hold an ambiguous or wide hole
        (counted) arrive with their ring points, values and hole cells already cut out (tpiv_postval_compact) and go to
        the triangulation -- in worker processes when fill_workers > 0: the jobs are handed out here and the answers read
        by _post_complete, so whatever the caller does in between (the next launch, the batch before) overlaps them.
        Returns the state _post_complete takes."""
        done, host, keep_alive = ticket
        done.synchronize()
        cnt = host["counts"].numpy().astype(np.int64)                # [n, 4] holes, ring, ambiguous, general
        nr, nc = host["fu"].shape[1:]
        n = cnt.shape[0]
        ring = cnt[:, 1]
        no_ring = ring == 0                  # nothing to interpolate from (B:300-304; the clean-pair quirk)
        too_many = ~no_ring & (4 * ring >= nr * nc)                  # points.size >= mask.size / 2 (B:299, 305)
        keep = ~no_ring & ~too_many
        need_host = keep & ((cnt[:, 2] + cnt[:, 3]) > 0)
        st = self.stats
        st["pairs"] += n
        st["dropped_no_invalid"] += int(no_ring.sum())
        st["dropped_too_many"] += int(too_many.sum())
        st["device_complete"] += int((keep & ~need_host).sum())
        st["host_fallback_diamonds_only"] += int((need_host & (cnt[:, 3] == 0)).sum())
        st["host_fallback_wide_holes"] += int((need_host & (cnt[:, 3] > 0)).sum())
        for _ in range(2 * int(too_many.sum())):                  # once for u, once for v (B:306, B:889-890)
            print(TOO_MANY_MSG)
        state = {"keep": keep, "need": np.flatnonzero(need_host), "host": host, "dev": keep_alive[0]}
        if state["need"].size:
            off = host["offsets"].numpy()                            # [2, n + 1]: ring / hole list starts per pair
            rc, uv, hc = host["ring_rc"].numpy(), host["ring_uv"].numpy(), host["hole_rc"].numpy()
            tot_r, tot_h = int(off[0, n]), int(off[1, n])
            if tot_r > rc.shape[0] or tot_h > hc.shape[0]:
                # heavily damaged fields: the lists are longer than the asynchronous copy carried -- fetch them whole
                ring_rc, ring_uv, hole_rc = keep_alive[1:4]
                rc, uv, hc = ring_rc[:tot_r].cpu().numpy(), ring_uv[:tot_r].cpu().numpy(), hole_rc[:tot_h].cpu().numpy()
            # (the lists list pair after pair, row-major inside a pair: np.argwhere's order, the reference's)
            jobs = [(rc[off[0, k]:off[0, k + 1]], uv[off[0, k]:off[0, k + 1]], hc[off[1, k]:off[1, k + 1]])
                    for k in state["need"]]
            state["holes"] = [j[2] for j in jobs]
            pool = self._fill_pool()
            if pool is not None:
                # (round 5: own worker processes behind pipes.  multiprocessing.Pool.map was synchronous here -- with map_async
                #  the pool's handler threads compete with this thread for the GIL and the whole host side got slower,
                #  6.5 k -> 4.8 k pairs/s -- and a batch of 64 pairs waited 2-4 ms for its triangulations)
                state["pending"] = (pool, pool.submit(jobs))
            else:
                if not getattr(OfflinePIV, "_blas_limited", False):
                    from ._qhull import blas_one_thread
                    OfflinePIV._blas_ctx = blas_one_thread
                    OfflinePIV._blas_limited = True
                with OfflinePIV._blas_ctx():
                    state["sols"] = qhull_fill_many(jobs)
        return state

    def _post_complete(self, state):
        """Second stage: the triangulation's values go -- flipped and scaled with the reference's expressions, cell by
        cell -- into the finished fields (and into the raw ones when they were asked for).  Returns the state with the
        per-pair keep flags (False: dropped)."""
        keep, need, host = state["keep"].copy(), state["need"], state["host"]
        if "pending" in state:
            pool, t_ = state.pop("pending")
            state["sols"] = pool.collect(t_, forget_older=True)
        if need.size:
            st = self.stats
            fu, fv = host["fu"].numpy(), host["fv"].numpy()
            uk, vk = (host["u"].numpy(), host["v"].numpy()) if "u" in host else (None, None)
            nr = fu.shape[1]
            st["host_fallback"] += len(state["sols"])
            ii, cells_l, vals_l = [], [], []
            for k, vals_k in enumerate(state["sols"]):
                if vals_k is None:
                    st["dropped_by_qhull"] += 1
                    keep[int(need[k])] = False
                    continue
                cells = state["holes"][k]
                ii.append(np.full(cells.shape[0], int(need[k]), dtype=np.int64))
                cells_l.append(cells)
                vals_l.append(vals_k)
            if ii:
                # ONE indexed assignment per field for the whole batch (round 5; per pair before: 4-6 fancy-index stores each)
                ii = np.concatenate(ii)
                cells = np.concatenate(cells_l).astype(np.int64, copy=False)
                vals = np.concatenate(vals_l)
                rr, cc = cells[:, 0], cells[:, 1]
                if uk is not None:
                    uk[ii, rr, cc] = vals[:, 0]
                    vk[ii, rr, cc] = vals[:, 1]
                pu = vals[:, 0] * self._scale / self._dt * 1000          # the reference's expression, cell by cell (B:896-898)
                pv = -vals[:, 1] * self._scale / self._dt * 1000
                fr = nr - 1 - rr
                fu[ii, fr, cc] = pu
                fv[ii, fr, cc] = pv
                if self.device_out:
                    # the same patches into the device copies of the finished fields: ONE small upload + index_put per batch
                    dev = state["dev"]["fu"].device
                    idx = torch.from_numpy(np.stack([ii, fr, cc])).to(dev)
                    val = torch.from_numpy(np.stack([pu, pv])).to(dev)
                    state["dev"]["fu"][idx[0], idx[1], idx[2]] = val[0]
                    state["dev"]["fv"][idx[0], idx[1], idx[2]] = val[1]
        state["keep_final"] = keep
        return state

    def _post_collect(self, ticket):
        """Both host stages at once; per pair None (dropped) or the raw (u, v) before the flip / scaling (the ticket
        must come from _post_submit(..., want_raw=True))."""
        state = self._post_complete(self._post_extract(ticket))
        uk, vk = state["host"]["u"].numpy(), state["host"]["v"].numpy()
        return [(uk[i], vk[i]) if state["keep_final"][i] else None for i in range(uk.shape[0])]

    def _post_validate_batch(self, u, v, inv):
        return self._post_collect(self._post_submit(u, v, inv, want_raw=True))

    def _post_pipeline(self, x, y, depth=1):
        """The host side of batched() as a pipeline: push(meta, ticket) after every launch returns the finished
        entries [(meta, per-pair results)] of the batch pushed `depth` + 1 launches before -- while the GPU works on batch k,
        the census of batch k - depth is taken and its triangulations go to the worker processes; their answers are read,
        patched in and handed out one push later.  The file path uses
        depth 2: a batch's upload (longer than its passes) then overlaps the passes of the batch before instead of being
        waited for.  flush() drains."""
        waiting, extracted = [], []

        def step(drain=False):
            # the census of the next batch first: it waits for the GPU, and behind that wait its triangulations go to the
            # workers; then the batch before it, whose triangulations were handed out one step ago and have had that whole
            # step (the wait included) to finish, is patched and handed out
            out = []
            if waiting:
                meta, ticket = waiting.pop(0)
                extracted.append((meta, self._post_extract(ticket) if ticket is not None else None))
            if extracted and (len(extracted) > 1 or (drain and not waiting)):
                meta, state = extracted.pop(0)
                out.append((meta, self._finish_batch(self._post_complete(state), x, y) if state is not None else []))
            return out

        class Pipe:
            def push(_, meta, ticket):
                out = step() if len(waiting) >= depth else []
                waiting.append((meta, ticket))
                return out

            def flush(_):
                out = []
                while waiting or extracted:
                    out += step(drain=True)
                return out
        return Pipe()

    def _finish_batch(self, state, x, y):
        """The finished tuples of a batch: the device has flipped and scaled the fields (_post_submit), the host stage has
        patched the filled cells (_post_complete); what is left is ONE copy of the two stacks out of the pinned staging
        memory (so that results a caller keeps do not pin pages) and the per-pair views.  Returns per pair None or
        (x, y, u, v); x, y are one pair of read-only arrays per batch (the reference makes fresh copies per pair; a caller
        that wants to write into them copies first)."""
        keep = state["keep_final"]
        if not keep.any():
            return [None] * keep.size
        if self.device_out:
            U, V = state["dev"]["fu"], state["dev"]["fv"]           # rows of the batch's device stacks (views)
        else:
            U, V = np.array(state["host"]["fu"].numpy()), np.array(state["host"]["fv"].numpy())
        xs, ys = x * self._scale, y * self._scale
        xs.flags.writeable = False
        ys.flags.writeable = False
        return [(xs, ys, U[k], V[k]) if keep[k] else None for k in range(keep.size)]

    def _finish(self, uv, x, y):
        """Flip and unit scaling of B:894-898 (numpy, the reference's own expressions)."""
        if uv is None:
            return None
        u, v = uv
        u = np.flip(u, axis=0)
        v = -np.flip(v, axis=0)
        u = u * self._scale / self._dt * 1000
        v = v * self._scale / self._dt * 1000
        return x * self._scale, y * self._scale, u, v

    # pairs per launch of __call__ (extension): the generator of the reference's API reads ahead and runs
    # `call_batch` pairs through batched(); the fields it yields, their order and the dropped pairs are those of
    # the one-pair-per-launch loop (call_batch = 1, the reference's B:868-901 literally)
    call_batch = 32

    def _one(self, i):
        """Pair i alone: decode on the host, one launch per pass, post-validation, flip / scale (B:868-898).
        None for an undecodable or dropped pair.  Plans of this path are kept per frame shape."""
        a, b = self._dataset[i]
        if a is None or b is None:
            return None
        a = a.to(self._device, non_blocking=True)
        b = b.to(self._device, non_blocking=True)
        shape = (int(a.shape[-2]), int(a.shape[-1]))
        plans = self._single_plans
        plan = plans.get(shape)
        if plan is None:
            plan = plans[shape] = engine.Plan(shape[0], shape[1], int(self._wind_size), int(self._overlap),
                                              n_pass=max(1, int(self._iter)), mode=self._mode,
                                              pass_scale=self._iter_scale, max_batch=1, device=self._device,
                                              val_ratio=self._val_ratio, val_win=self._val_win,
                                              precision=self._precision)
        u, v, inv = plan.run(a, b)
        w, o, _, _ = plan.geometry[-1]
        x, y = get_coordinates(shape, w, o)
        return self._finish(self._post_validate_batch(u, v, inv)[0], x, y)

    def __call__(self) -> Generator:
        if int(self.call_batch) > 1 and len(self._dataset) > 1:
            # the reference hands out FRESH x, y per pair (B:899-900: x * scale makes a new array), which a caller may
            # write into; batched() shares one read-only pair of coordinate arrays per batch, so copy here
            for _, x, y, u, v in self.batched(int(self.call_batch)):
                yield x.copy(), y.copy(), u, v
            return
        end_time = time()
        for i in range(len(self._dataset)):
            if self.verbose:
                print(f"Load time {(time() - end_time):.3f} sec", end=" ")
            start = time()
            out = self._one(i)
            if out is None:
                continue
            yield out
            end_time = time()
            if self.verbose:
                print(f"Batch finished in {(end_time - start):.3f} sec")

    def batched(self, batch_size: int = 32, indices=None) -> Generator:
        """Like __call__, but reads, uploads and processes `batch_size` pairs per launch.  Native reader
        threads (io.ReadAhead) put the next batches' files into pinned staging memory -- uncompressed BMPs as their RAW
        FILE BYTES (no host decode: header skip, row flip, padding strip and palette / gray conversion
        run on the device, tpiv_bmp_unpack), other formats decoded on the host -- while the GPU works on
        the current batch (triple-buffered staging, uploads on their own stream).  Yields (pair_index, x, y, u, v); dropped pairs yield nothing."""
        from .io import ReadAhead, parse_bmp_headers, stage_raw
        idx = list(range(len(self._dataset))) if indices is None else list(indices)
        if not idx:
            return
        first = None
        for i in idx:                       # frame shape from the first decodable pair
            a0, _ = self._dataset[i]
            if a0 is not None:
                first = (i, tuple(a0.shape))
                break
        if first is None:
            return
        H, W = first[1]
        plan = self._get_plan(H, W, max_batch=batch_size)
        import os as _os
        # one staging slot per file: the largest of the first pair's files (a run's files share one format)
        # and a headerless frame, rounded up to 4 KiB
        sizes = [H * W] + [_os.path.getsize(p_) for p_ in self._dataset.img_pairs[first[0]] if _os.path.exists(p_)]
        cap = (max(sizes) + 4095) // 4096 * 4096
        # (page-locking half a gigabyte takes a tenth of a second: the staging buffers are kept for the next call)
        key = (batch_size, cap)
        prev = getattr(self, "_reader", None)            # a reader that outlived its (abandoned) generator still fills the buffers
        if prev is not None:
            prev.close()
            # ... and uploads / unpack kernels of that run may still be queued on ITS upload stream and on the compute
            # stream: the staging and device buffers below are the same memory, and the new run's first upload carries no
            # dependency on them (an old upload landing afterwards would be unpacked as the new batch 0)
            torch.cuda.synchronize(self._device)
        if getattr(self, "_stage_key", None) != key:
            # three staging buffers: one being read into, one uploading, one of slack
            self._stage = [torch.empty(2 * batch_size, cap, dtype=torch.uint8).pin_memory() for _ in range(3)]
            self._stage_key = key
        stage = self._stage
        pairs = self._dataset.img_pairs
        paths = []
        for i in idx:                       # slot 2k: frame a of the batch's pair k, slot 2k + 1: frame b
            paths += [pairs[i][0], pairs[i][-1]]
        # the loader is native (tpiv_reader_*): reader threads of the library stream the run's files into the staging
        # buffers ahead of this loop, which only blocks -- without the GIL -- for the next complete batch.  Formats the
        # device cannot unpack are not read here at all: they take the per-file path (host decode) below
        rd = ReadAhead(paths, 2 * batch_size, [t.data_ptr() for t in stage], cap, threads=self.read_threads,
                       read=str(paths[0]).lower().endswith(".bmp"))
        self._reader = rd
        decoders = None
        w, o, _, _ = plan.geometry[-1]
        x, y = get_coordinates((H, W), w, o)
        dev = self._device
        pipe = self._post_pipeline(x, y, depth=self.pipeline_depth)

        def emit(finished):
            """Results of finished batches in dataset order; the pairs that were not staged run now."""
            for (order, chunk), res in finished:
                res = iter(res)
                for i, staged in order:
                    out = next(res) if staged else self._one(i)
                    if out is not None:
                        if self.device_out and not staged:      # the one-pair path finishes on the host
                            out = out[:2] + tuple(torch.from_numpy(np.ascontiguousarray(f)).to(dev) for f in out[2:])
                        yield (i,) + out

        # uploads run on their own stream into a double-buffered device copy of the staging slots, so that the PCIe
        # transfer of batch n + 1 (4 MP: 268 MB, ~5 ms) overlaps the passes of batch n (~3 ms) instead of preceding them
        cur = torch.cuda.current_stream(dev)
        up_stream = torch.cuda.Stream(dev)
        key_d = (batch_size, cap, str(dev))
        if getattr(self, "_raw_dev_key", None) != key_d:
            self._raw_dev = [torch.empty(2 * batch_size, cap, dtype=torch.uint8, device=dev) for _ in range(2)]
            self._raw_dev_key = key_d
        raw_dev = self._raw_dev
        consumed = [None, None]             # event: the unpack kernel that read raw_dev[k] has run
        release = None                      # upload event of the batch before (its staging buffer is still held)
        n_up = 0

        def let_go(rel):
            if rel is not None:
                if rel[0] is not None:
                    rel[0].synchronize()              # its upload is through: the readers may refill the staging buffer
                rd.release()

        try:
            for s0 in range(0, len(idx), batch_size):
                got = rd.next()
                if got is None:
                    break
                buf, sizes = got
                ids = idx[s0:s0 + batch_size]
                raw = stage[buf].numpy()
                lays = parse_bmp_headers(raw[:len(sizes)], sizes, H, W)
                rest = [j for j, lay in enumerate(lays) if lay is None]
                if rest:
                    if decoders is None:
                        from concurrent.futures import ThreadPoolExecutor
                        decoders = ThreadPoolExecutor(max_workers=self.read_threads)
                    for j, lay in zip(rest, decoders.map(lambda j: stage_raw(paths[2 * s0 + j], raw[j], H, W), rest)):
                        lays[j] = lay
                chunk, desc_a, desc_b, lut_a, lut_b, order = [], [], [], [], [], []
                for k, i in enumerate(ids):
                    la, lb = lays[2 * k], lays[2 * k + 1]
                    order.append((i, la is not None and lb is not None))
                    if la is None or lb is None:
                        # not stageable (undecodable, or a frame shape other than the batch's): the pair takes the
                        # one-pair path when its turn comes -- which skips an undecodable pair like B:138-139 and
                        # gives another shape its own plan
                        continue
                    desc_a.append([2 * k * cap, la[0], la[1], la[2], la[3], 0])
                    desc_b.append([(2 * k + 1) * cap, lb[0], lb[1], lb[2], lb[3], 0])
                    lut_a.append(la[4])
                    lut_b.append(lb[4])
                    chunk.append(i)
                ticket, up = None, None
                if chunk:
                    n, n_slots = len(chunk), len(ids)
                    dbuf, n_up = n_up % 2, n_up + 1
                    with torch.cuda.stream(up_stream):
                        if consumed[dbuf] is not None:
                            up_stream.wait_event(consumed[dbuf])
                        raw_d = raw_dev[dbuf][:2 * n_slots]
                        raw_d.copy_(stage[buf][:2 * n_slots], non_blocking=True)
                        up = torch.cuda.Event()
                        up.record(up_stream)
                    # unpacked frame order: every a of the batch, then every b (two contiguous stacks)
                    desc_d = torch.tensor(desc_a + desc_b, dtype=torch.int64).to(dev, non_blocking=True)
                    lut_d = torch.from_numpy(np.stack(lut_a + lut_b)).to(dev, non_blocking=True)
                    cur.wait_event(up)
                    frames = engine.bmp_unpack(raw_d.view(-1), desc_d, lut_d, H, W)      # [2n, H, W]: a_0..a_n-1, b_0..b_n-1
                    consumed[dbuf] = torch.cuda.Event()
                    consumed[dbuf].record(cur)
                    u, v, inv = plan.run(frames[:n], frames[n:])
                    ticket = self._post_submit(u, v, inv)
                let_go(release)
                release = (up,)
                # the host work of the PREVIOUS batches runs while the GPU works on this one
                yield from emit(pipe.push((order, chunk), ticket))
            let_go(release)
            release = None
            yield from emit(pipe.flush())
        finally:
            # consumer finished, raised, or abandoned the generator (GeneratorExit lands here): stop the reader threads,
            # then wait for every upload / unpack still queued (they read the staging buffers and write raw_dev, which the
            # next batched() call on this object reuses with a fresh upload stream and no events to order itself behind)
            rd.close()
            torch.cuda.synchronize(dev)
            if decoders is not None:
                decoders.shutdown(wait=False)


class ResidentPIV(OfflinePIV):
    """OfflinePIV over frame pairs that already live on the GPU (uint8 tensors [n, H, W]): the same
    passes, post-validation, flip and scaling, without dataset / decoding / upload.  Extension used by
    the end-to-end benchmark and by callers that acquire straight into device memory."""

    def __init__(self, frames_a: torch.Tensor, frames_b: torch.Tensor, wind_size: int, overlap: int,
                 multipass: int = 1, multipass_mode: str = "CWS", dt: int = 1, scale: float = 1.,
                 multipass_scale: float = 2., precision: str = "exact", validation_ratio: float = 1.2,
                 validation_window: int = 3) -> None:
        if frames_a.shape != frames_b.shape or frames_a.dim() != 3 or frames_a.dtype != torch.uint8 \
                or frames_b.dtype != torch.uint8:
            raise ValueError("ResidentPIV: two uint8 tensors [n, H, W] of one shape")
        if precision not in PRECISIONS:
            raise KeyError(precision)
        self._precision = precision
        self._val_ratio, self._val_win = float(validation_ratio), int(validation_window)
        self._wind_size, self._overlap, self._dt = wind_size, overlap, dt
        self._iter, self._iter_scale, self._scale = multipass, multipass_scale, scale
        self._device = _require_gpu(frames_a.device)
        self._iter_function = IterModMap.functions[multipass_mode]
        self._mode = multipass_mode
        self._plan = None
        self._A, self._B = frames_a.contiguous(), frames_b.contiguous()
        self._dataset = range(frames_a.shape[0])
        self.reset_stats()

    def frame_shape(self):
        return tuple(self._A.shape[1:]) if len(self) else None

    # launches in flight before a batch's census is read.  Two: the copies of a batch's results run on a stream of their own,
    # as kernels (rocprofv3 shows __amd_rocclr_copyBuffer), and the passes of the NEXT batch leave them no registers on any CU
    # until they end -- with one launch in flight the host got a batch's census when the GPU had just run dry (round 5: GPU
    # timeline of the 'isolated spots' case, 0.8 ms idle per 64 pairs)
    resident_depth = 2

    def batched(self, batch_size: int = 32, indices=None) -> Generator:
        idx = list(range(len(self))) if indices is None else list(indices)
        if not idx:
            return
        H, W = self._A.shape[1:]
        plan = self._get_plan(H, W, max_batch=batch_size)
        w, o, _, _ = plan.geometry[-1]
        x, y = get_coordinates((H, W), w, o)
        pipe = self._post_pipeline(x, y, depth=self.resident_depth)

        def emit(finished):
            for chunk_ids, res in finished:
                for i, out in zip(chunk_ids, res):
                    if out is not None:
                        yield (i,) + out

        for s in range(0, len(idx), batch_size):
            chunk = idx[s:s + batch_size]
            # a run of consecutive pairs is a view of the resident frames; anything else is gathered (a copy of 2 x 4 MB per pair:
            # 0.36 ms per 64 pairs at 4 MP -- the check is per launch, so a stream that repeats or skips stays copy-free per run)
            if chunk[-1] - chunk[0] == len(chunk) - 1 and chunk == list(range(chunk[0], chunk[0] + len(chunk))):
                A, B = self._A[chunk[0]:chunk[0] + len(chunk)], self._B[chunk[0]:chunk[0] + len(chunk)]
            else:
                sel = torch.tensor(chunk, device=self._device)
                A, B = self._A.index_select(0, sel), self._B.index_select(0, sel)
            u, v, inv = plan.run(A, B)
            # host work of the previous batches overlaps this batch's kernels
            yield from emit(pipe.push(chunk, self._post_submit(u, v, inv)))
        yield from emit(pipe.flush())

    def __call__(self) -> Generator:
        for _, x, y, u, v in self.batched(1):
            yield x.copy(), y.copy(), u, v


class OnlinePIV:
    """The reference's live-acquisition class is a constructor-only stub (B:906-927: it stores its arguments
    and resolves the device; there is no processing method).  Mirrored as such -- frames acquired straight
    into GPU memory go through ResidentPIV."""

    def __init__(self, folder: str, device: str, file_fmt: str, wind_size: int, overlap: int, iterations: int = 1,
                 dt: int = 1, scale: float = 1., resize: int = 2, iter_scale: float = 2.) -> None:
        self._wind_size = wind_size
        self._overlap = overlap
        self._dt = dt
        self._iter = iterations
        self._iter_scale = iter_scale
        self._resize = resize
        self._scale = scale
        self._device = DeviceMap.devicies[device]           # KeyError like B:927
