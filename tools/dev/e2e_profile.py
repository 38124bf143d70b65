"""Where does the resident generator spend its host time?  (development aid: wraps the stages with timers)"""
import os, sys, time, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import torchpiv_amd as T
from torchpiv_amd import backend as B, engine
import e2e_generator as E

T_ = collections.defaultdict(float)
N_ = collections.defaultdict(int)


def timed(obj, name, label=None):
    f = getattr(obj, name)
    label = label or name

    def w(*a, **k):
        t = time.perf_counter()
        r = f(*a, **k)
        T_[label] += time.perf_counter() - t
        N_[label] += 1
        return r
    setattr(obj, name, w)


if __name__ == "__main__":
    n, workers = int(sys.argv[1]) if len(sys.argv) > 1 else 256, int(sys.argv[2]) if len(sys.argv) > 2 else 8
    BATCH = int(sys.argv[3]) if len(sys.argv) > 3 else 64
    KIND = sys.argv[4] if len(sys.argv) > 4 else "spots"
    DISTINCT = min(n, 128)
    A, Bf = E.make_frames(DISTINCT, 2048, 2048, KIND)
    order = [i % DISTINCT for i in range(n)]
    piv = T.ResidentPIV(A, Bf, 64, 32, multipass=2, multipass_mode="CWS")
    piv.fill_workers = workers
    if len(sys.argv) > 5:
        piv.resident_depth = int(sys.argv[5])
    sum(1 for _ in piv.batched(BATCH, indices=order))
    timed(piv, "_post_submit")
    if getattr(piv, "_plan", None) is not None:
        timed(piv._plan, "run", "plan.run (launch)")
    timed(piv, "_post_extract")
    timed(piv, "_post_complete")
    timed(piv, "_finish_batch")
    timed(engine, "postval")
    real_fill = B.qhull_fill_many
    pool = piv._fill_pool()
    if pool is not None:
        timed(pool, "submit", "workers.submit")
        timed(pool, "collect", "workers.collect")
    ev_sync = torch.cuda.Event.synchronize

    def sync(self):
        t = time.perf_counter()
        ev_sync(self)
        T_["event.synchronize"] += time.perf_counter() - t
    torch.cuda.Event.synchronize = sync
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    k = sum(1 for _ in piv.batched(BATCH, indices=order))
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f"{n / dt:.0f} pairs/s, {dt * 1e3 / (n / BATCH):.2f} ms per batch of {BATCH}, yielded {k}")
    for key in T_:
        print(f"  {key:20s} {T_[key] * 1e3 / (n / BATCH):7.2f} ms per batch  ({N_[key]} calls)")
    piv.close()
