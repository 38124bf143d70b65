"""Multi-GPU layer: image pairs are independent (PIVbackend.py:868-901 carries no state from
one pair to the next), so they shard across ranks with no data-path collective; the only
communication is ONE gather of the finished (u, v) fields at the end of the stream.

One process per GPU, torch.distributed with backend "nccl" (= RCCL over xGMI on ROCm) on the
GPU box and "gloo" in the CPU tests.  Dropped pairs (the reference's hole-fill quirk) make the
shard sizes data dependent, so a small count exchange precedes the payload.
"""
from __future__ import annotations

import os

import torch
import torch.distributed as dist


def init_from_env(backend: str | None = None):
    """Initialise torch.distributed from RANK / WORLD_SIZE / MASTER_* (torchrun).  Returns
    (rank, world, local_rank).  No-op for single-process runs."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        if backend is None:
            # TPIV_DIST_BACKEND=gloo: rehearsal of the multi-rank path on a box with fewer GPUs than
            # ranks (RCCL refuses two ranks on one device); payloads are then staged through the host
            backend = os.environ.get("TPIV_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if backend == "nccl":
            torch.cuda.set_device(local)
            dist.init_process_group(backend, rank=rank, world_size=world,
                                    device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
    return rank, world, local


def shard_indices(n_pairs: int, rank: int, world: int, policy: str = "block"):
    """Pair indices owned by `rank`.  'block': contiguous balanced runs of n // world pairs, the first
    n % world ranks taking one more (keeps file reads sequential per rank; a shard is empty only when
    n < world); 'cyclic': i % world == rank (balances a stream of unknown length)."""
    if policy == "cyclic":
        return list(range(rank, n_pairs, world))
    if policy != "block":
        raise KeyError(policy)
    q, r = divmod(n_pairs, world)
    lo = rank * q + min(rank, r)
    return list(range(lo, lo + q + (1 if rank < r else 0)))


def _numel(shape) -> int:
    n = 1
    for d in shape:
        n *= int(d)
    return n


def gather_fields(ids: torch.Tensor, fields: torch.Tensor, dst: int = 0, group=None, stats: dict = None):
    """Gather variable-length shards onto rank `dst`.

    ids    int64 [n_local]            dataset index of every field this rank produced
    fields       [n_local, C, R, S]   e.g. C = 2 for (u, v); same dtype/shape tail on all ranks
    Returns (ids_all, fields_all) sorted by dataset index on `dst`, (None, None) elsewhere.
    Ranks with nothing to send pass fields of shape [0, ...]; their tail shape is taken from the ranks
    that have data (exchanged with the counts), so an empty shard needs no plan.
    stats (optional dict) receives what a scaling run needs to be read: the bytes this rank put into the payload collective
    (`payload_bytes`, padded to the largest shard), the bytes of it that are fields and ids (`useful_bytes`), the shard
    sizes of all ranks (`counts`) and the number of collectives issued.
    TWO collectives: one count / shape all-gather (56 bytes per rank) + one payload gather onto `dst` whose
    byte buffer carries the fields AND their ids: on xGMI's full mesh the seven shards arrive over seven
    different links at once, and no other rank has to hold the whole result (an all-gather would move and
    store world x more).
    """
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        if stats is not None:
            stats.update(payload_bytes=0, useful_bytes=0, counts=[int(ids.numel())], collectives=0)
        order = torch.argsort(ids)
        return ids[order], fields[order]
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    out_dev = fields.device
    if dist.get_backend(group) == "gloo" and fields.is_cuda:      # gloo moves host memory only
        fields = fields.cpu()
        ids = ids.cpu()
    dev = fields.device
    # counts and tail shapes in one small exchange: [n, ndim_tail, d0, d1, d2, d3, d4]
    tail_local = tuple(fields.shape[1:])
    if len(tail_local) > 5:
        raise ValueError("gather_fields: at most 5 trailing dimensions")
    meta = torch.zeros(7, dtype=torch.int64, device=dev)
    meta[0] = ids.numel()
    meta[1] = len(tail_local)
    for k, d in enumerate(tail_local):
        meta[2 + k] = d
    metas = torch.zeros(world * 7, dtype=torch.int64, device=dev)
    dist.all_gather_into_tensor(metas, meta, group=group)
    metas = metas.view(world, 7).cpu()
    counts = metas[:, 0]
    n_max = int(counts.max().item())
    tail = tail_local
    have = (counts > 0).nonzero().flatten()
    if have.numel():
        first = metas[int(have[0])]
        tail = tuple(int(t) for t in first[2:2 + int(first[1])])
        for r_ in have.tolist():
            t_r = tuple(int(t) for t in metas[r_][2:2 + int(metas[r_][1])])
            if t_r != tail:
                raise ValueError(f"gather_fields: rank {r_} holds fields of shape {t_r}, rank {int(have[0])} {tail}")
    if ids.numel() == 0:
        fields = fields.new_zeros((0,) + tail)
    if stats is not None:
        stats.update(payload_bytes=0, useful_bytes=0, counts=[int(c) for c in counts.tolist()], collectives=1)
    if n_max == 0:          # nothing anywhere (every rank knows it from the counts: no second collective)
        if rank != dst:
            return None, None
        return ids.new_zeros(0).to(out_dev), fields.new_zeros((0,) + tail).to(out_dev)
    # ONE payload collective: the ids ride behind the fields in the same byte buffer (fields padded to the
    # largest shard -- dist.gather needs equal sizes; block shards differ by at most one pair unless pairs
    # were dropped)
    fields = fields.contiguous()
    f_bytes = n_max * _numel(tail) * fields.element_size()
    buf = torch.zeros(f_bytes + n_max * 8, dtype=torch.uint8, device=dev)
    pad_i = torch.full((n_max,), -1, dtype=torch.int64, device=dev)
    pad_i[: ids.numel()] = ids.to(dev)
    if ids.numel():
        buf[: fields.numel() * fields.element_size()] = fields.reshape(-1).view(torch.uint8)
    buf[f_bytes:] = pad_i.view(torch.uint8)
    if rank == dst:
        all_b = torch.empty(world, buf.numel(), dtype=torch.uint8, device=dev)
        lst = list(all_b.unbind(0))
    else:
        lst = None
    if stats is not None:
        stats.update(payload_bytes=int(buf.numel()), collectives=2,
                     useful_bytes=int(ids.numel()) * (_numel(tail) * fields.element_size() + 8))
    dst_global = dist.get_global_rank(group, dst) if group is not None else dst
    # (gather is implemented by both backends this package runs on -- RCCL and gloo; any error here is a
    #  real one and propagates: a fallback decided per rank could leave the ranks in different collectives)
    dist.gather(buf, gather_list=lst, dst=dst_global, group=group)
    if rank != dst:
        return None, None
    all_f = all_b[:, :f_bytes].contiguous().view(fields.dtype).view((world * n_max,) + tail)
    all_i = all_b[:, f_bytes:].contiguous().view(torch.int64).reshape(-1)
    keep = all_i >= 0
    all_i, all_f = all_i[keep], all_f[keep]
    order = torch.argsort(all_i)
    return all_i[order].to(out_dev), all_f[order].to(out_dev)


def run_sharded(piv, batch_size: int = 32, policy: str = "block", group=None):
    """Process an OfflinePIV dataset across all ranks.  Every rank runs its shard through
    piv.batched(); rank 0 returns (ids, x, y, uv[n, 2, R, S]) for the pairs that survived,
    in dataset order; other ranks return None."""
    import numpy as np
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    mine = shard_indices(len(piv), rank, world, policy)
    ids, uv, xy = [], [], None
    # the finished fields stay where the last kernel left them: batched() hands out rows of the per-batch device stacks
    # (hole fills of the host stage scattered in), and the gather below reads device memory -- nothing goes host -> device
    was = piv.device_out
    piv.device_out = True
    try:
        for i, x, y, u, v in piv.batched(batch_size, indices=mine):
            ids.append(i)
            uv.append(torch.stack([u, v]))
            xy = (x, y)
    finally:
        piv.device_out = was
    dev = piv._device
    if uv:
        f = torch.stack(uv)
    else:       # empty shard (or every pair of it dropped): gather_fields takes the grid from the other ranks
        f = torch.zeros((0, 2, 0, 0), dtype=torch.float64, device=dev)
    i_all, f_all = gather_fields(torch.tensor(ids, dtype=torch.int64, device=dev), f, group=group)
    if rank != 0:
        return None
    if xy is None and len(piv):        # rank 0 itself yielded nothing: the grid depends on the geometry only
        from . import backend
        w, o = int(piv._wind_size), int(piv._overlap)
        for _ in range(max(1, int(piv._iter)) - 1):
            w, o = int(w // piv._iter_scale), int(o // piv._iter_scale)
        shape = piv.frame_shape()
        if shape is not None:
            x, y = backend.get_coordinates(shape, w, o)
            xy = (x * piv._scale, y * piv._scale)
    return i_all.cpu().numpy(), xy, f_all.cpu().numpy()
