"""Headless job runner: what the reference's GUI worker does around the generator
(workers.py:29-124, PIVWorker.run) without Qt -- drive OfflinePIV over a folder, optionally export
every pair, accumulate the ensemble statistics (mean, Reynolds stresses, gradients, vorticity and
shear of the mean field) and write the statistics table.

Export formats are the reference's (PlotterFunctions.py:16-65): `save_table` writes a
comma-separated text table with a header line and `%.6f` numbers, `save_binary` a `.npy` stack of
the dict's arrays; existing files are not overwritten but get " (n)" appended (`uniquify`).

The statistics are accumulated as streaming sums in float64 (the reference stacks every field and
takes two-pass means; the results agree to rounding), so a 4000-pair run needs O(1) memory, and the
sums reduce across ranks with ONE all-reduce for multi-GPU runs.
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
    """Streaming first and second moments of (u, v) fields."""

    def __init__(self):
        self.n = 0
        self.su = self.sv = self.suu = self.svv = self.suv = None

    def add(self, u: np.ndarray, v: np.ndarray):
        u = u.astype(np.float64)
        v = v.astype(np.float64)
        if self.su is None:
            z = np.zeros_like(u)
            self.su, self.sv, self.suu, self.svv, self.suv = z.copy(), z.copy(), z.copy(), z.copy(), z.copy()
        self.n += 1
        self.su += u
        self.sv += v
        self.suu += u * u
        self.svv += v * v
        self.suv += u * v

    def allreduce(self, device=None, group=None):
        """Sum the accumulators over all ranks (one collective)."""
        import torch
        import torch.distributed as dist
        if not dist.is_initialized() or dist.get_world_size(group) == 1:
            return
        shape = None if self.su is None else self.su.shape
        shapes = [None] * dist.get_world_size(group)
        dist.all_gather_object(shapes, shape, group=group)
        shape = next((s for s in shapes if s is not None), None)
        if shape is None:
            return
        if self.su is None:
            z = np.zeros(shape)
            self.su, self.sv, self.suu, self.svv, self.suv = z.copy(), z.copy(), z.copy(), z.copy(), z.copy()
        pack = np.concatenate([np.array([float(self.n)]), self.su.ravel(), self.sv.ravel(), self.suu.ravel(),
                               self.svv.ravel(), self.suv.ravel()])
        dev = device if (device is not None and dist.get_backend(group) == "nccl") else "cpu"
        t = torch.from_numpy(pack).to(dev)
        dist.all_reduce(t, group=group)
        pack = t.cpu().numpy()
        k = int(np.prod(shape))
        self.n = int(round(pack[0]))
        parts = [pack[1 + i * k: 1 + (i + 1) * k].reshape(shape) for i in range(5)]
        self.su, self.sv, self.suu, self.svv, self.suv = parts

    def table(self, x: np.ndarray, y: np.ndarray) -> dict:
        """The statistics table of workers.py:85-118 (same keys, same order, same gradient call:
        np.gradient(avg, dx, dy, edge_order=2) with dx, dy taken at the grid centre in metres)."""
        n = max(self.n, 1)
        avg_u, avg_v = self.su / n, self.sv / n
        uu = self.suu / n - avg_u * avg_u
        vv = self.svv / n - avg_v * avg_v
        uv = self.suv / n - avg_u * avg_v
        mid_i, mid_j = x.shape[-2] // 2, x.shape[-1] // 2
        dx = (x[mid_i, mid_j + 1] - x[mid_i, mid_j]) / 1000
        dy = (y[mid_i + 1, mid_j] - y[mid_i, mid_j]) / 1000
        dUy, dUx = np.gradient(avg_u, dx, dy, edge_order=2)
        dVy, dVx = np.gradient(avg_v, dx, dy, edge_order=2)
        return {
            "x[mm]": x,
            "y[mm]": y,
            "Vx[m/s]": avg_u,
            "Vy[m/s]": avg_v,
            "(vx-Vx)(vy-Vy)[m^2/s^2]": uv,
            "(vx-Vx)^2[m^2/s^2]": uu,
            "(vy-Vy)^2[m^2/s^2]": vv,
            "dVx/dx[1/s]": dUx,
            "dVx/dy[1/s]": dUy,
            "dVy/dx[1/s]": dVx,
            "dVy/dy[1/s]": dVy,
            "W[1/s]": (dVx - dUy),
            "S[1/s]": (dVx + dUy),
        }


def run_folder(folder: str, device: str, file_fmt: str, wind_size: int, overlap: int, multipass: int = 1,
               multipass_mode: str = "CWS", dt: int = 1, scale: float = 1.0, multipass_scale: float = 2.0,
               folder_mode: str = "pairs", save_opt: str = "Dont save", save_dir: str = "Out",
               batch_size: int = 32, on_pair=None, distributed: bool = False):
    """Process a folder like PIVWorker.run.  save_opt: "Dont save" | "Save all binary" |
    "Save all text" | "Save statistics" (anything but "Dont save" also writes the statistics table).
    Returns (table, n_pairs_done); with distributed=True every rank processes its shard of the
    pairs and rank 0 returns the table of the whole ensemble (other ranks: (None, n_local))."""
    piv = OfflinePIV(folder, device, file_fmt, wind_size, overlap, multipass=multipass,
                     multipass_mode=multipass_mode, dt=dt, scale=scale, multipass_scale=multipass_scale,
                     folder_mode=folder_mode)
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
    stats = EnsembleStats()
    x = y = None
    done = 0
    for i, xx, yy, u, v in piv.batched(batch_size, indices=indices):
        x, y = xx, yy
        stats.add(u, v)
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
        stats.allreduce(device=piv._device)
        if x is None:                       # a rank whose shard yielded nothing still needs the grid
            from .backend import get_coordinates
            plan = piv._plan
            if plan is not None:
                w, o, _, _ = plan.geometry[-1]
                x, y = get_coordinates((plan.H, plan.W), w, o)
                x, y = x * scale, y * scale
    if rank != 0:
        return None, done
    if stats.su is None or x is None:
        return None, done
    table = stats.table(x, y)
    if save_opt != "Dont save":
        save_table(f"{name}_statistics.txt", save_dir, table.copy())
    return table, done
