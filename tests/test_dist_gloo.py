"""The N > 1 path on CPU: two processes, gloo backend -- sharding policy and the single
end-of-stream gather with data-dependent shard sizes (dropped pairs)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank),
                      WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    from torchpiv_amd import dist as pdist
    r, w, _ = pdist.init_from_env(backend="gloo")
    assert (r, w) == (rank, world)
    n_pairs = 11
    mine = pdist.shard_indices(n_pairs, rank, world, "block")
    # rank 1 "drops" one of its pairs, as the hole-fill quirk would
    kept = [i for i in mine if not (rank == 1 and i == mine[1])]
    fields = torch.stack([torch.full((2, 3, 4), float(i), dtype=torch.float64) for i in kept]) \
        if kept else torch.zeros(0, 2, 3, 4, dtype=torch.float64)
    ids, allf = pdist.gather_fields(torch.tensor(kept, dtype=torch.int64), fields)
    if rank == 0:
        q.put((ids.tolist(), allf[:, 0, 0, 0].tolist()))
    else:
        assert ids is None and allf is None
    # fewer pairs than ranks: rank 1's shard is empty and it has no plan to take the grid from --
    # its placeholder is [0, 2, 0, 0]; the tail shape travels with the counts
    mine1 = pdist.shard_indices(1, rank, world, "block")
    assert len(mine1) == (1 if rank == 0 else 0)
    f1 = torch.full((1, 2, 3, 4), 7.0, dtype=torch.float64) if mine1 else torch.zeros(0, 2, 0, 0, dtype=torch.float64)
    ids1, all1 = pdist.gather_fields(torch.tensor(mine1, dtype=torch.int64), f1)
    if rank == 0:
        assert ids1.tolist() == [0] and tuple(all1.shape) == (1, 2, 3, 4) and float(all1.sum()) == 7.0 * 24
    # ... and the other way round (the destination rank is the empty one)
    f2 = torch.full((2, 2, 3, 4), 5.0, dtype=torch.float64) if rank == 1 else torch.zeros(0, 2, 0, 0, dtype=torch.float64)
    ids2, all2 = pdist.gather_fields(torch.tensor([4, 3] if rank == 1 else [], dtype=torch.int64), f2)
    if rank == 0:
        assert ids2.tolist() == [3, 4] and tuple(all2.shape) == (2, 2, 3, 4)
    # cyclic policy partitions too
    cyc = pdist.shard_indices(n_pairs, rank, world, "cyclic")
    t = torch.zeros(n_pairs)
    t[cyc] = 1
    dist.all_reduce(t)
    assert bool((t == 1).all())
    # ensemble statistics: the ranks' fields gathered onto rank 0 give the single-process table bit for bit
    from torchpiv_amd.runner import EnsembleStats
    rng = np.random.default_rng(11)
    fields = [(rng.standard_normal((4, 5)), rng.standard_normal((4, 5))) for _ in range(6)]
    st = EnsembleStats()
    for k, (u, v) in enumerate(fields):
        if k % world == rank:
            st.add(u, v, index=k)
    st.gather()
    if rank == 0:
        full = EnsembleStats()
        for u, v in fields:
            full.add(u, v)
        assert st.n == 6 and all(np.array_equal(a, b) for a, b in zip(st.moments(), full.moments()))
    else:
        assert st.n == 0
    # streaming mode: per-rank running accumulators, ONE payload gather of six planes per rank, Chan merge on rank 0 --
    # the two-pass moments to rounding (uneven shards; with world 2 rank 1 holds 2 of 7 fields, offsets keep it honest)
    fields7 = [(rng.standard_normal((4, 5)) * 3 + 100.0, rng.standard_normal((4, 5)) - 50.0) for _ in range(7)]
    ss = EnsembleStats(streaming=True)
    for k, (u, v) in enumerate(fields7):
        if (k % 3 == 1) == (rank == 1):
            ss.add(u, v, index=k)
    calls = []
    real_ag, real_g = dist.all_gather_into_tensor, dist.gather
    dist.all_gather_into_tensor = lambda *a, **k: (calls.append("all_gather"), real_ag(*a, **k))[1]
    dist.gather = lambda *a, **k: (calls.append("gather"), real_g(*a, **k))[1]
    ss.gather()
    dist.all_gather_into_tensor, dist.gather = real_ag, real_g
    assert calls == ["all_gather", "gather"], calls
    if rank == 0:
        full = EnsembleStats()
        for u, v in fields7:
            full.add(u, v)
        assert ss.n == 7
        for a, b in zip(ss.moments(), full.moments()):
            assert np.abs(a - b).max() <= 1e-12 * 100.0, float(np.abs(a - b).max())
    else:
        assert ss.n == 0
    dist.barrier()
    dist.destroy_process_group()


def _worker4(rank, world, port, q):
    """Four ranks; rank 2 drops EVERY pair of its shard (an all-dropped rank), rank 3's shard is cut short; then
    the all-empty stream.  Counts the collectives a gather issues: one meta all-gather + one payload gather."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank),
                      WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    from torchpiv_amd import dist as pdist
    pdist.init_from_env(backend="gloo")
    calls = []
    real_ag, real_g = dist.all_gather_into_tensor, dist.gather
    dist.all_gather_into_tensor = lambda *a, **k: (calls.append("all_gather"), real_ag(*a, **k))[1]
    dist.gather = lambda *a, **k: (calls.append("gather"), real_g(*a, **k))[1]
    n_pairs = 13
    mine = pdist.shard_indices(n_pairs, rank, world, "block")            # 4, 3, 3, 3
    kept = [] if rank == 2 else (mine[:-1] if rank == 3 else mine)
    fields = torch.stack([torch.full((2, 5, 3), float(i), dtype=torch.float64) for i in kept]) \
        if kept else torch.zeros(0, 2, 0, 0, dtype=torch.float64)
    ids, allf = pdist.gather_fields(torch.tensor(kept, dtype=torch.int64), fields)
    assert calls == ["all_gather", "gather"], calls
    if rank == 0:
        assert tuple(allf.shape[1:]) == (2, 5, 3)
        q.put((ids.tolist(), allf[:, 1, 4, 2].tolist()))
    else:
        assert ids is None and allf is None
    # nothing anywhere: every rank learns it from the counts, no payload collective
    del calls[:]
    ids0, f0 = pdist.gather_fields(torch.zeros(0, dtype=torch.int64), torch.zeros(0, 2, 0, 0, dtype=torch.float64))
    assert calls == ["all_gather"], calls
    if rank == 0:
        assert ids0.numel() == 0 and f0.shape[0] == 0
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(240)
def test_gather_four_ranks_one_all_dropped():
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker4, args=(r, 4, port, q)) for r in range(4)]
    for p in procs:
        p.start()
    ids, vals = q.get(timeout=180)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    # 13 pairs: rank 0 owns 0..3, rank 1 4..6, rank 2 7..9 (all dropped), rank 3 10..12 (12 dropped)
    assert ids == [0, 1, 2, 3, 4, 5, 6, 10, 11]
    assert vals == [float(i) for i in ids]


def _worker8(rank, world, port, q):
    """Eight ranks (the node the scaling run uses) and five pairs: ranks 5..7 own nothing (n < world), rank 1 drops its
    only pair; the `stats` of the gather -- what bench.py puts on its line -- are checked on every rank."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank),
                      WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    from torchpiv_amd import dist as pdist
    pdist.init_from_env(backend="gloo")
    n_pairs = 5
    mine = pdist.shard_indices(n_pairs, rank, world, "block")            # 1, 1, 1, 1, 1, 0, 0, 0
    assert len(mine) == (1 if rank < 5 else 0)
    kept = [] if rank == 1 else mine
    fields = torch.stack([torch.full((2, 6, 7), float(i) + 0.5, dtype=torch.float64) for i in kept]) \
        if kept else torch.zeros(0, 2, 0, 0, dtype=torch.float64)
    st = {}
    ids, allf = pdist.gather_fields(torch.tensor(kept, dtype=torch.int64), fields, stats=st)
    assert st["collectives"] == 2 and st["counts"] == [1, 0, 1, 1, 1, 0, 0, 0]
    assert st["payload_bytes"] == 2 * 6 * 7 * 8 + 8                      # every rank ships the largest shard's size
    assert st["useful_bytes"] == len(kept) * (2 * 6 * 7 * 8 + 8)
    if rank == 0:
        assert tuple(allf.shape) == (4, 2, 6, 7)
        q.put((ids.tolist(), allf[:, 1, 5, 6].tolist()))
    else:
        assert ids is None and allf is None
    # a second gather on the same group (config 2 gathers in every step): same result, same statistics
    st2 = {}
    ids2, _ = pdist.gather_fields(torch.tensor(kept, dtype=torch.int64), fields, stats=st2)
    assert st2 == st and (rank != 0 or ids2.tolist() == ids.tolist())
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_gather_eight_ranks_three_empty_one_all_dropped():
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker8, args=(r, 8, port, q)) for r in range(8)]
    for p in procs:
        p.start()
    ids, vals = q.get(timeout=240)
    for p in procs:
        p.join(90)
        assert p.exitcode == 0
    assert ids == [0, 2, 3, 4] and vals == [i + 0.5 for i in ids]


@pytest.mark.timeout(180)
def test_gather_two_ranks():
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    ids, vals = q.get(timeout=120)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    # 11 pairs, balanced blocks: rank 0 owns 0..5, rank 1 owns 6..10 and dropped pair 7
    assert ids == [0, 1, 2, 3, 4, 5, 6, 8, 9, 10]
    assert vals == [float(i) for i in ids]


def test_shard_indices_cover_everything():
    from torchpiv_amd import dist as pdist
    for n in (0, 1, 7, 8, 4000):
        for world in (1, 2, 8):
            for pol in ("block", "cyclic"):
                got = sorted(sum((pdist.shard_indices(n, r, world, pol) for r in range(world)), []))
                assert got == list(range(n))
    assert len(pdist.shard_indices(4000, 3, 8)) == 500
    # balanced: 9 pairs on 8 ranks leave no rank empty, sizes differ by at most one
    sizes = [len(pdist.shard_indices(9, r, 8)) for r in range(8)]
    assert sizes == [2, 1, 1, 1, 1, 1, 1, 1]
    assert [len(pdist.shard_indices(3, r, 8)) for r in range(8)] == [1, 1, 1, 0, 0, 0, 0, 0]
