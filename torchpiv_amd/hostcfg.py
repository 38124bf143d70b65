"""Host-side budget of a rank: how many cores this process may count on, and the reader / fill-worker counts that
follow from them.

The hot path runs on the GPU, but the generator (OfflinePIV.batched) keeps host threads busy beside it: reader
threads that stream files into pinned staging memory (native, torchpiv_amd/csrc/c_api.cpp tpiv_reader_*), worker
PROCESSES that run the Delaunay hole fill of the pairs the device cannot finish (PIVbackend.py:284-308 through
Qhull), and the main thread.  One process per GPU means eight such sets per node: the counts must come from the
cores a rank really has -- its affinity mask and the container's CPU quota, divided by the ranks of the node --
not from a constant (round 3 started 8 + 8 per rank whatever the node looked like).
"""
from __future__ import annotations

import os


def cgroup_cpu_quota():
    """CPU quota of the container in cores (cgroup v2 cpu.max, v1 cfs quota), None when unlimited / unknown."""
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:
            q, p = f.read().split()[:2]
        if q != "max":
            return float(q) / float(p)
    except (OSError, ValueError):
        pass
    try:
        with open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us") as f:
            q = int(f.read())
        with open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as f:
            p = int(f.read())
        if q > 0 and p > 0:
            return q / p
    except (OSError, ValueError):
        pass
    return None


def cores_available() -> float:
    """Cores this process may use: the affinity mask, capped by the container's CPU quota."""
    try:
        n = float(len(os.sched_getaffinity(0)))
    except (AttributeError, OSError):
        n = float(os.cpu_count() or 1)
    q = cgroup_cpu_quota()
    return min(n, q) if q else n


def local_world() -> int:
    """Ranks that share this node (torchrun: LOCAL_WORLD_SIZE; else WORLD_SIZE; else 1)."""
    for k in ("LOCAL_WORLD_SIZE", "WORLD_SIZE"):
        v = os.environ.get(k)
        if v and v.isdigit() and int(v) > 0:
            return int(v)
    return 1


def host_budget(ranks_on_node: int | None = None, cores: float | None = None) -> dict:
    """{'cores': cores of the node share, 'per_rank': cores per rank, 'read_threads', 'fill_workers'}.

    One core is the main thread's (Python: census, patching, yield).  The rest is split between the reader threads
    (a page-cache read into pinned memory runs at ~3 GB/s per thread; the PCIe link takes 57 GB/s = 6 750 4-MP pairs/s
    per GPU, i.e. more than 8 threads never pay) and the fill workers (0.1 ms of Qhull per pair and worker: 8 workers
    keep up with 10 k pairs/s).  With fewer than 3 cores per rank the triangulations run in the main thread."""
    ranks = ranks_on_node if ranks_on_node else local_world()
    total = cores if cores is not None else cores_available()
    per = max(1.0, total / max(1, ranks))
    spare = max(0, int(per) - 1)
    read_threads = max(1, min(8, (spare + 1) // 2))
    fill_workers = max(0, min(8, spare - read_threads))
    return {"cores": total, "ranks_on_node": ranks, "per_rank": per, "read_threads": read_threads,
            "fill_workers": fill_workers}


def pin_rank(local_rank: int, ranks_on_node: int | None = None) -> list | None:
    """Give this rank its own slice of the affinity mask (call before the first GPU call and before worker processes
    are started: children inherit it).  Only when the mask really is the budget: under a CPU-time quota smaller than the
    mask (a container limited by cpu.max) slicing the mask would confine a rank to a few hardware threads that other
    tenants may be using, so the mask is left alone and only the thread counts are sized.  Returns the cores set."""
    ranks = ranks_on_node if ranks_on_node else local_world()
    try:
        mask = sorted(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        return None
    q = cgroup_cpu_quota()
    if ranks <= 1 or len(mask) < 2 * ranks or (q is not None and q < len(mask)):
        return None
    per = len(mask) // ranks
    mine = mask[local_rank * per:(local_rank + 1) * per]
    os.sched_setaffinity(0, mine)
    return mine


def tree_cpu_seconds() -> float:
    """user + system CPU seconds of this process, its threads and its live child processes (the fill workers)."""
    import psutil
    me = psutil.Process()
    t = me.cpu_times()
    total = t.user + t.system
    for ch in me.children(recursive=True):
        try:
            c = ch.cpu_times()
            total += c.user + c.system
        except psutil.Error:
            pass
    return total
