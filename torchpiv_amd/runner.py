"""Headless job runner: what the reference's GUI worker does around the generator
(workers.py:29-124, PIVWorker.run) without Qt -- drive OfflinePIV over a folder, optionally export
every pair, accumulate the ensemble statistics (mean, Reynolds stresses, gradients, vorticity and
shear of the mean field) and write the statistics table.

Export formats are the reference's (PlotterFunctions.py:16-65): `save_table` writes a
comma-separated text table with a header line and `%.6f` numbers, `save_binary` a `.npy` stack of
the dict's arrays; existing files are not overwritten but get " (n)" appended (`uniquify`).

The statistics follow the reference's two-pass arithmetic on the stacked fields bit for bit (see
EnsembleStats); multi-rank runs gather the fields onto rank 0 with the single end-of-run gather.
"""
from __future__ import annotations

import itertools
import os

import numpy as np

from .backend import OfflinePIV

KEYS_PAIR = ("x[mm]", "y[mm]", "Vx[m/s]", "Vy[m/s]")


def _numbered(path: str):
    """path, then 'stem (1).ext', 'stem (2).ext', ... -- the naming scheme of PlotterFunctions.py:16-24."""
    stem, ext = os.path.splitext(path)
    yield path
    for n in itertools.count(1):
        yield f"{stem} ({n}){ext}"


def uniquify(path: str) -> str:
    """First name of the numbered series that does not exist yet (PlotterFunctions.py:16-24)."""
    return next(p for p in _numbered(path) if not os.path.exists(p))


def _reserve(path: str) -> str:
    """Like uniquify, but the name is CLAIMED atomically (O_EXCL), so that two processes exporting into
    one directory can never pick the same file (the reference is single-process and only tests)."""
    for cand in _numbered(path):
        try:
            os.close(os.open(cand, os.O_CREAT | os.O_EXCL | os.O_WRONLY, 0o644))
            return cand
        except FileExistsError:
            continue


def save_binary(name, path, data: dict, sep: str = ", "):
    """PlotterFunctions.py:48-53: one .npy holding the dict's arrays stacked along a new first axis, in
    dict order.  Returns the path written."""
    os.makedirs(path, exist_ok=True)
    target = _reserve(os.path.join(path, name))
    with open(target, "wb") as f:
        np.save(f, np.stack([np.asarray(v) for v in data.values()], axis=0))
    return target


def save_table(name, path, data: dict, sep: str = ", "):
    """PlotterFunctions.py:55-65: a text table, one column per key (arrays flattened row-major), the
    keys joined by `sep` as header line, numbers as '%.6f'.  Returns the path written."""
    os.makedirs(path, exist_ok=True)
    table = np.column_stack([np.ravel(np.asarray(v)) for v in data.values()])
    target = _reserve(os.path.join(path, name))
    np.savetxt(target, table, fmt="%.6f", delimiter=sep, header=sep.join(data), comments="")
    return target


class EnsembleStats:
    """Ensemble statistics of workers.py:85-118 -- mean, Reynolds stresses, gradients, vorticity, shear --
    with the reference's arithmetic: it stacks every field and takes TWO-PASS means with numpy, which
    accumulates along the stack axis in order, so `mean = ((f0 + f1) + f2 ...) / n` and
    `uu = sum_k (f_k - mean)^2 / n` in that order are reproduced bit for bit (a streaming
    E[x^2] - E[x]^2 would cancel where the two-pass form does not).  The fields are kept on the host until
    the end of the run (O(n) memory, see moments()) and the moments come from a kernel on the GPU
    (tpiv_ensemble_moments: one thread per grid cell walks the stack in order), chunk of cells by chunk of
    cells, or from numpy.

    streaming=True (extension for long runs): nothing is kept per pair.  Five float64 accumulators per grid cell --
    mean u, mean v and the centred sums M2_uu, M2_vv, C_uv -- are updated field by field with Welford's recurrence
    (no E[x^2] - E[x]^2 cancellation), every rank accumulates its own shard, and the ranks' accumulators are merged on
    rank 0 with Chan's pairwise formula after ONE payload gather of 6 planes per rank (the five accumulators and the
    count).  A 4000 x 1023 x 1023 run needs 50 MB per rank instead of 67 GB on rank 0; the moments agree with the
    two-pass mode to rounding (<= 1e-12 relative to the field's scale, tests/test_runner.py), not bit for bit."""

    def __init__(self, streaming: bool = False):
        self.streaming = bool(streaming)
        self.ids, self.u, self.v = [], [], []
        self._n = 0
        self._acc = None                 # streaming: float64 [5, R, S] = mean u, mean v, M2_uu, M2_vv, C_uv

    @property
    def n(self):
        return self._n if self.streaming else len(self.u)

    def add(self, u: np.ndarray, v: np.ndarray, index=None):
        if self.streaming:
            u = np.asarray(u, dtype=np.float64)
            v = np.asarray(v, dtype=np.float64)
            if self._acc is None:
                self._acc = np.zeros((5,) + u.shape, dtype=np.float64)
            self._n += 1
            a = self._acc
            du, dv = u - a[0], v - a[1]                  # against the OLD means
            a[0] += du / self._n
            a[1] += dv / self._n
            du2, dv2 = u - a[0], v - a[1]                # against the NEW means
            a[2] += du * du2
            a[3] += dv * dv2
            a[4] += du * dv2
            return
        self.ids.append(len(self.ids) if index is None else int(index))
        self.u.append(np.asarray(u, dtype=np.float64))
        self.v.append(np.asarray(v, dtype=np.float64))

    @staticmethod
    def _merge(na, A, nb, B):
        """Chan et al.: accumulators of two disjoint samples -> accumulators of their union."""
        if nb == 0:
            return na, A
        if na == 0:
            return nb, B
        n = na + nb
        du, dv = B[0] - A[0], B[1] - A[1]
        w = na * nb / n
        out = np.empty_like(A)
        out[0] = A[0] + du * (nb / n)
        out[1] = A[1] + dv * (nb / n)
        out[2] = A[2] + B[2] + du * du * w
        out[3] = A[3] + B[3] + dv * dv * w
        out[4] = A[4] + B[4] + du * dv * w
        return n, out

    def gather(self, device=None, group=None):
        """Multi-rank runs: bring every rank's fields onto rank 0, in dataset order (one count exchange
        + one padded gather, torchpiv_amd.dist.gather_fields); other ranks end up empty."""
        import torch
        import torch.distributed as dist
        from . import dist as pdist
        if not dist.is_initialized() or dist.get_world_size(group) == 1:
            return
        dev = device if (device is not None and dist.get_backend(group) == "nccl") else "cpu"
        if self.streaming:
            # six planes per rank: the five accumulators and the count (ranks without a field send nothing and take the
            # grid from the others, like an empty shard of fields)
            if self._n:
                f = torch.from_numpy(np.concatenate([self._acc, np.full((1,) + self._acc.shape[1:], float(self._n))])[None]).to(dev)
            else:
                f = torch.zeros((0, 6, 0, 0), dtype=torch.float64, device=dev)
            me = dist.get_rank(group)
            ids, planes = pdist.gather_fields(torch.tensor([me] if self._n else [], dtype=torch.int64, device=dev), f, group=group)
            self._n, self._acc = 0, None
            if ids is None:
                return
            planes = planes.cpu().numpy()
            for k in range(planes.shape[0]):             # rank order
                self._n, self._acc = self._merge(self._n, self._acc, int(round(planes[k, 5].flat[0])), planes[k, :5].copy())
            return
        f = torch.from_numpy(np.stack([np.stack(self.u), np.stack(self.v)], axis=1)).to(dev) if self.u \
            else torch.zeros((0, 2, 0, 0), dtype=torch.float64, device=dev)
        ids, fields = pdist.gather_fields(torch.tensor(self.ids, dtype=torch.int64, device=dev), f, group=group)
        if ids is None:
            self.ids, self.u, self.v = [], [], []
            return
        fields = fields.cpu().numpy()
        self.ids = ids.cpu().tolist()
        self.u = [fields[k, 0] for k in range(fields.shape[0])]
        self.v = [fields[k, 1] for k in range(fields.shape[0])]

    # cells per chunk of moments(): bounds the stacked copy (host) and its upload (device) to
    # 2 fields x n x CHUNK_BYTES-worth of cells, whatever the length of the run
    CHUNK_BYTES = 256 << 20

    def moments(self, device=None):
        """(avg_u, avg_v, uu, vv, uv) in dataset order of the fields.

        The per-pair fields themselves stay in host memory until the end of the run (the reference's
        arithmetic is two-pass over the whole stack: 16 bytes per vector and pair, e.g. 4000 pairs of 127 x 127
        = 1 GB, of 1023 x 1023 = 67 GB).  Everything on top of that is bounded: the stack is cut into chunks of
        grid cells (CHUNK_BYTES per field component), each chunk is stacked, uploaded and reduced on its own
        (tpiv_ensemble_moments walks a cell's stack in order, so cutting along the cells changes no bit), and
        a chunk that does not fit the GPU is reduced by numpy instead."""
        if self.streaming:
            a = self._acc
            return a[0].copy(), a[1].copy(), a[2] / self._n, a[3] / self._n, a[4] / self._n
        order = np.argsort(np.asarray(self.ids), kind="stable")
        n = len(order)
        shape = self.u[order[0]].shape
        cells = int(np.prod(shape))
        step = max(1, min(cells, self.CHUNK_BYTES // (8 * n)))
        out = np.empty((5, cells), dtype=np.float64)
        flat_u = [self.u[k].reshape(-1) for k in order]
        flat_v = [self.v[k].reshape(-1) for k in order]
        for c0 in range(0, cells, step):
            c1 = min(cells, c0 + step)
            U = np.stack([f[c0:c1] for f in flat_u])
            V = np.stack([f[c0:c1] for f in flat_v])
            res = None
            if device is not None:
                import torch
                from . import engine
                try:
                    res = [o.cpu().numpy() for o in engine.ensemble_moments(torch.from_numpy(U).to(device),
                                                                            torch.from_numpy(V).to(device))]
                except torch.OutOfMemoryError:
                    torch.cuda.empty_cache()
                    res = None
            if res is None:
                avg_u = np.mean(U, axis=0, dtype=np.float64)
                avg_v = np.mean(V, axis=0, dtype=np.float64)
                du, dv = U - avg_u, V - avg_v
                res = [avg_u, avg_v, np.mean(du ** 2, axis=0, dtype=np.float64), np.mean(dv ** 2, axis=0, dtype=np.float64),
                       np.mean(du * dv, axis=0, dtype=np.float64)]
            for q in range(5):
                out[q, c0:c1] = res[q]
        return tuple(out[q].reshape(shape) for q in range(5))

    def table(self, x: np.ndarray, y: np.ndarray, device=None) -> dict:
        """The statistics table of workers.py:85-118: same keys, same order.  The gradients are
        np.gradient(mean, step_x, step_y, edge_order=2) like the reference -- including its argument order
        (the x step is handed to the ROW axis) and its choice of the grid steps at the centre cell, in
        metres (the coordinates are millimetres)."""
        avg_u, avg_v, uu, vv, uv = self.moments(device)
        cy, cx = x.shape[-2] // 2, x.shape[-1] // 2
        step_x = (x[cy, cx + 1] - x[cy, cx]) / 1000
        step_y = (y[cy + 1, cx] - y[cy, cx]) / 1000
        u_rows, u_cols = np.gradient(avg_u, step_x, step_y, edge_order=2)       # d/d(row), d/d(column)
        v_rows, v_cols = np.gradient(avg_v, step_x, step_y, edge_order=2)
        names = ("x[mm]", "y[mm]", "Vx[m/s]", "Vy[m/s]", "(vx-Vx)(vy-Vy)[m^2/s^2]", "(vx-Vx)^2[m^2/s^2]",
                 "(vy-Vy)^2[m^2/s^2]", "dVx/dx[1/s]", "dVx/dy[1/s]", "dVy/dx[1/s]", "dVy/dy[1/s]", "W[1/s]", "S[1/s]")
        cols = (x, y, avg_u, avg_v, uv, uu, vv, u_cols, u_rows, v_cols, v_rows, v_cols - u_rows, v_cols + u_rows)
        return dict(zip(names, cols))


def run_folder(folder: str, device: str, file_fmt: str, wind_size: int, overlap: int, multipass: int = 1,
               multipass_mode: str = "CWS", dt: int = 1, scale: float = 1.0, multipass_scale: float = 2.0,
               folder_mode: str = "pairs", save_opt: str = "Dont save", save_dir: str = "Out",
               batch_size: int = 32, on_pair=None, distributed: bool = False, stats_on_device: bool = True,
               precision: str = "exact", streaming_stats: bool = False):
    """Process a folder like PIVWorker.run.  save_opt: "Dont save" | "Save all binary" |
    "Save all text" | "Save statistics" (anything but "Dont save" also writes the statistics table).
    streaming_stats: running accumulators instead of the stacked fields (EnsembleStats(streaming=True): O(1) memory in
    the number of pairs, one 6-plane gather between ranks; moments to rounding instead of bit for bit).
    Returns (table, n_pairs_done); with distributed=True every rank processes its shard of the
    pairs and rank 0 returns the table of the whole ensemble (other ranks: (None, n_local))."""
    piv = OfflinePIV(folder, device, file_fmt, wind_size, overlap, multipass=multipass,
                     multipass_mode=multipass_mode, dt=dt, scale=scale, multipass_scale=multipass_scale,
                     folder_mode=folder_mode, precision=precision)
    if len(piv) == 0:
        return None, 0
    rank, world = 0, 1
    indices = None
    if distributed:
        import torch.distributed as dist
        from . import dist as pdist
        if dist.is_initialized():
            rank, world = dist.get_rank(), dist.get_world_size()
        indices = pdist.shard_indices(len(piv), rank, world)
    name = os.path.basename(os.path.normpath(folder))
    stats = EnsembleStats(streaming=streaming_stats)
    if distributed and world > 1:
        piv.auto_host_config()           # reader threads / fill workers from this rank's share of the node's cores
    x = y = None
    done = 0
    for i, xx, yy, u, v in piv.batched(batch_size, indices=indices):
        x, y = xx, yy
        stats.add(u, v, index=i)
        done += 1
        output = dict(zip(KEYS_PAIR, (x, y, u, v)))
        # single process: the reference's names (numbered in processing order); several ranks: the
        # dataset index goes into the name, so that files map to pairs whatever the interleaving
        tag = f"_pair_{i:06d}" if world > 1 else "_pair"
        if save_opt == "Save all binary":
            save_binary(f"{name}{tag}.npy", save_dir, output.copy())
        elif save_opt == "Save all text":
            save_table(f"{name}{tag}.txt", save_dir, output.copy())
        if on_pair is not None:
            on_pair(i, output)
    if distributed and world > 1:
        stats.gather(device=piv._device)
        if x is None and rank == 0:         # rank 0's own shard yielded nothing: the grid depends on the geometry only
            from .backend import get_coordinates
            shape = piv.frame_shape()
            if shape is not None:
                w, o = int(wind_size), int(overlap)
                for _ in range(max(1, int(multipass)) - 1):
                    w, o = int(w // multipass_scale), int(o // multipass_scale)
                x, y = get_coordinates(shape, w, o)
                x, y = x * scale, y * scale
    if rank != 0:
        return None, done
    if stats.n == 0 or x is None:
        return None, done
    table = stats.table(x, y, device=piv._device if stats_on_device else None)
    if save_opt != "Dont save":
        save_table(f"{name}_statistics.txt", save_dir, table.copy())
    return table, done
