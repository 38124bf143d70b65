"""The triangulation step of the reference's fillMissingValues (PIVbackend.py:300-302) as a pure function of
small arrays.  This module imports numpy and scipy ONLY: it is the target of OfflinePIV's worker processes
(`fill_workers`), which therefore never import torch, never load libtorchpiv_hip.so and never touch the GPU.
(The workers are started with the `spawn` method, which re-imports the caller's `__main__`: scripts that use
`fill_workers > 0` need the usual `if __name__ == "__main__":` guard.)

The reference calls scipy.interpolate.LinearNDInterpolator(points, values)(targets).  That class is Qhull's Delaunay
triangulation of the points plus, per target, a walk to the simplex that holds it and a barycentric sum -- but its
Python-level set-up costs 2-4 ms for the two dozen ring points of a typical pair, against 0.1-0.3 ms for the
triangulation itself.  `qhull_fill` runs the SAME triangulation (scipy.spatial.Delaunay: same Qhull, same options, same
point order -- and with it the same tie-break on the co-circular diamonds of a lattice) and evaluates it with the
arithmetic of scipy's _interpnd.pyx spelled out in numpy: bit-identical results (tests/test_host_logic.py compares the
two on random hole patterns), a tenth of the time."""
import numpy as np


def qhull_fill_reference(points, values, targets):
    """The reference's literal call (kept as the yardstick for the test)."""
    from scipy.interpolate import LinearNDInterpolator
    try:
        return LinearNDInterpolator(points, values)(targets)
    except Exception:
        return None


def _diamond_fill(tri, points, values, targets):
    """Every target an ISOLATED hole (its four lattice neighbours are ring points): the four are co-circular around it with no
    ring point inside, so they are one cell of the Delaunay subdivision and Qhull's 'Qt' splits that cell by ONE of its
    diagonals -- which one is its tie-break, a function of the whole point set.  The target is the midpoint of
    either diagonal, so whichever of the two triangles along the chosen diagonal the interpolator's walk ends in, the
    barycentric coordinates are (1/2, 1/2, 0) exactly (lattice coordinates: the 2 x 2 inverse has entries 0, +-1/2, +-1): the
    value is 0.5 a + 0.5 b of the diagonal's ends.  So only the DIAGONAL has to be read off the triangulation: no barycentric
    transforms (one LAPACK call per simplex), no simplex walk.  Returns None when a target is not such a hole or the diagonal
    cannot be read (the caller then takes the general path)."""
    pts = np.asarray(points, dtype=np.int64)
    tg = np.asarray(targets, dtype=np.int64)
    n = pts.shape[0]
    if pts.ndim != 2 or pts.shape[1] != 2 or tg.shape[0] == 0 or n > 2048:
        return None               # (thousands of ring points: wide holes are among them anyway -- the general path)
    # point index by lattice cell (dense table with a one-cell margin: the neighbour look-ups below need no range checks),
    # edges of the triangulation as a dense n x n table: on two dozen points numpy's set routines (isin, unique, searchsorted)
    # cost more than the triangulation itself
    r0, c0 = int(min(pts[:, 0].min(), tg[:, 0].min())) - 1, int(min(pts[:, 1].min(), tg[:, 1].min())) - 1
    R, K = int(max(pts[:, 0].max(), tg[:, 0].max())) - r0 + 2, int(max(pts[:, 1].max(), tg[:, 1].max())) - c0 + 2
    if R * K > (1 << 22):
        return None
    lut = np.full((R, K), -1, dtype=np.int64)
    lut[pts[:, 0] - r0, pts[:, 1] - c0] = np.arange(n)
    tr, tc = tg[:, 0] - r0, tg[:, 1] - c0
    iN, iS, iW, iE = lut[tr - 1, tc], lut[tr + 1, tc], lut[tr, tc - 1], lut[tr, tc + 1]
    if min(iN.min(), iS.min(), iW.min(), iE.min()) < 0:
        return None
    sp = tri.simplices
    adj = np.zeros((n, n), dtype=bool)
    for i, j in ((0, 1), (1, 2), (0, 2)):
        adj[sp[:, i], sp[:, j]] = True
        adj[sp[:, j], sp[:, i]] = True
    ns, ew = adj[iN, iS], adj[iW, iE]
    if not (ns ^ ew).all():
        return None
    a = np.where(ns, iN, iW)
    b = np.where(ns, iS, iE)
    return 0.5 * values[a] + 0.5 * values[b]


def qhull_fill(points, values, targets, diamonds=True):
    """Delaunay-linear interpolation of `values` [n, k] given at integer `points` [n, 2], evaluated at
    `targets` [m, 2].  None when Qhull refuses the points (the reference's bare `except` then drops the pair).
    diamonds: take the short cut for fields whose holes are all isolated (_diamond_fill: same bits, half the time)."""
    from scipy.spatial import Delaunay
    try:
        tri = Delaunay(points)                   # what LinearNDInterpolator builds (qhull.Delaunay(points))
    except Exception:
        return None
    x = np.ascontiguousarray(targets, dtype=np.float64)
    values = np.asarray(values, dtype=np.float64)
    flat = values.ndim == 1
    if flat:
        values = values[:, None]
    if diamonds:
        out = _diamond_fill(tri, points, values, targets)
        if out is not None:
            return out[:, 0] if flat else out
    # the simplex walk of LinearNDInterpolator._do_evaluate (same routine, same start-from-the-last-hit order)
    s = tri.find_simplex(x)
    inside = s >= 0
    T = tri.transform[np.where(inside, s, 0)]    # [m, 3, 2]: inverse edge matrix and the reference vertex
    d0 = x[:, 0] - T[:, 2, 0]
    d1 = x[:, 1] - T[:, 2, 1]
    # _barycentric_coordinates: c[i] = sum_j T[i, j] (x[j] - r[j]) accumulated from 0, c[2] = 1 - c[0] - c[1]
    c0 = (0.0 + T[:, 0, 0] * d0) + T[:, 0, 1] * d1
    c1 = (0.0 + T[:, 1, 0] * d0) + T[:, 1, 1] * d1
    c2 = (1.0 - c0) - c1
    simp = tri.simplices[np.where(inside, s, 0)]
    out = np.zeros((x.shape[0], values.shape[1]), dtype=np.float64)
    for j, c in enumerate((c0, c1, c2)):         # out += c[j] * values[vertex j], in vertex order
        out = out + c[:, None] * values[simp[:, j]]
    out[~inside] = np.nan                        # outside the hull: fill_value
    return out[:, 0] if flat else out


def single_thread_blas():
    """Worker initialiser (and, once, the main process when it triangulates itself): scipy computes the barycentric
    transforms of a triangulation with one LAPACK call per simplex, and a multi-threaded OpenBLAS synchronises its thread
    pool around every one of those 2 x 2 problems -- measured on a 16-core share of a 256-thread host: 1.0 ms of CPU per
    pair against 0.23 ms with one BLAS thread (same bits: the factorisation of a 2 x 2 matrix is not split over threads)."""
    import os
    for k in ("OPENBLAS_NUM_THREADS", "OMP_NUM_THREADS", "MKL_NUM_THREADS"):
        os.environ[k] = "1"
    try:
        from threadpoolctl import threadpool_limits
        global _BLAS_LIMIT
        _BLAS_LIMIT = threadpool_limits(limits=1, user_api="blas")         # kept alive: the limit lasts as long as the object
    except Exception:                                                      # noqa: BLE001 -- the env variables cover a fresh import
        pass


def blas_one_thread():
    """Context manager for the in-process triangulations: one BLAS thread for their duration only (the main process may
    want its threads back for other numpy work)."""
    try:
        from threadpoolctl import threadpool_limits
        return threadpool_limits(limits=1, user_api="blas")
    except Exception:                                                      # noqa: BLE001
        import contextlib
        return contextlib.nullcontext()


def qhull_fill_many(jobs):
    """[qhull_fill(*job) for job in jobs] -- one task of the worker pool carries several pairs (IPC per task, not per pair)."""
    return [qhull_fill(*job) for job in jobs]


def worker_loop(conn):
    """Body of a fill-worker process (FillWorkers below): lists of jobs in, lists of qhull_fill results out, one message at a
    time, until the pipe closes or a None arrives."""
    single_thread_blas()
    while True:
        try:
            jobs = conn.recv()
        except (EOFError, OSError):
            break
        if jobs is None:
            break
        conn.send(qhull_fill_many(jobs))
    conn.close()


def _job_bytes(part):
    return sum(a.nbytes for job in part for a in job)


class FillWorkers:
    """`n` worker processes behind one duplex pipe each.  Unlike multiprocessing.Pool there are no handler threads in the
    calling process (they compete with it for the GIL: an asynchronous Pool.map made the whole generator slower) and the
    exchange is split: submit(jobs) hands the jobs of a batch out and returns at once, collect(ticket) reads the answers --
    the caller launches the next batch and finishes the previous one in between (backend.OfflinePIV._post_pipeline).
    Messages beyond LIGHT bytes could fill a pipe while its reader is itself blocked writing, so a batch that holds one is
    exchanged on the spot, one message per worker at a time (send to all, then read from all: no cycle of blocked writers)."""
    LIGHT = 24 * 1024            # bytes per message that may stay in flight (two batches outstanding: well inside a socket buffer)

    def __init__(self, n):
        import multiprocessing as mp
        ctx = mp.get_context("spawn")   # the workers never see the caller's HIP state
        self.conns, self.procs = [], []
        for _ in range(n):
            mine, theirs = ctx.Pipe(duplex=True)
            p = ctx.Process(target=worker_loop, args=(theirs,), daemon=True)
            p.start()
            theirs.close()
            self.conns.append(mine)
            self.procs.append(p)
        self.n = n
        self._order, self._parts, self._ready = [], {}, {}      # tickets in flight (oldest first), their message counts, answers read early

    def _recv(self, w):
        try:
            return self.conns[w].recv()
        except (EOFError, OSError) as e:
            raise RuntimeError(f"fill worker {w} died (exit code {self.procs[w].exitcode})") from e

    def _drain(self, upto=None):
        """Read the answers of the outstanding tickets, oldest first (every pipe is first in, first out), up to ticket `upto`."""
        while self._order and (upto is None or upto in self._order):
            t = self._order.pop(0)
            self._ready[t] = [s for w in range(self._parts.pop(t)) for s in self._recv(w)]

    def submit(self, jobs):
        """Ticket for collect(): the jobs cut into one run per worker (the order of the jobs is the order of the answers)."""
        t = self._next = getattr(self, "_next", 0) + 1
        per = max(1, -(-len(jobs) // self.n))
        parts = [jobs[k:k + per] for k in range(0, len(jobs), per)]
        if all(_job_bytes(p_) <= self.LIGHT for p_ in parts):
            for w, p_ in enumerate(parts):
                self.conns[w].send(p_)
            self._order.append(t)
            self._parts[t] = len(parts)
            return t
        # a heavy batch: what is still in flight is read first (its answers sit in front in every pipe), then single jobs are
        # exchanged now, one message per worker and round
        self._drain()
        out = []
        for k in range(0, len(jobs), self.n):
            rnd = jobs[k:k + self.n]
            for w, j in enumerate(rnd):
                self.conns[w].send([j])
            for w in range(len(rnd)):
                out += self._recv(w)
        self._ready[t] = out
        return t

    def collect(self, ticket, forget_older=False):
        """The answers of `ticket`.  forget_older: answers of earlier tickets that nobody collected (a generator abandoned with
        batches in flight) are dropped -- for callers that collect in the order they submit."""
        if ticket not in self._ready:
            self._drain(ticket)
        if forget_older:
            for t in [t for t in self._ready if t < ticket]:
                del self._ready[t]
        return self._ready.pop(ticket)

    def terminate(self):
        for c in self.conns:
            try:
                c.send(None)
                c.close()
            except (OSError, ValueError, BrokenPipeError):
                pass
        for p in self.procs:
            p.join(timeout=1.0)
            if p.is_alive():
                p.terminate()
        self.conns, self.procs = [], []
