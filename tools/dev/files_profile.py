"""Where does the file path (OfflinePIV.batched over 8-bit BMPs) spend its time?  (development aid)"""
import os, sys, time, collections, tempfile, shutil
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import torchpiv_amd as T
from torchpiv_amd import backend as B, engine, io as pio
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
    n, reps = 128, int(sys.argv[1]) if len(sys.argv) > 1 else 8
    workers = int(sys.argv[2]) if len(sys.argv) > 2 else 8
    threads = int(sys.argv[3]) if len(sys.argv) > 3 else 8
    kind = sys.argv[4] if len(sys.argv) > 4 else "spots"
    depth = int(sys.argv[5]) if len(sys.argv) > 5 else 2
    batch = int(sys.argv[6]) if len(sys.argv) > 6 else 32
    from PIL import Image
    A, Bf = E.make_frames(n, 2048, 2048, kind)
    d = tempfile.mkdtemp()
    for i in range(n):
        Image.fromarray(A[i].cpu().numpy(), "L").save(os.path.join(d, f"img{i:05d}_a.bmp"))
        Image.fromarray(Bf[i].cpu().numpy(), "L").save(os.path.join(d, f"img{i:05d}_b.bmp"))
        for r_ in range(1, reps):
            for s_ in "ab":
                os.link(os.path.join(d, f"img{i:05d}_{s_}.bmp"), os.path.join(d, f"img{i + r_ * n:05d}_{s_}.bmp"))
    del A, Bf
    piv = T.OfflinePIV(d, "cuda:0", "bmp", 64, 32, multipass=2, multipass_mode="CWS")
    piv.fill_workers, piv.read_threads, piv.pipeline_depth = workers, threads, depth
    sum(1 for _ in piv.batched(batch, indices=range(2 * batch)))
    timed(pio.ReadAhead, "next", "reader.next (wait)")
    timed(pio, "parse_bmp_headers")
    timed(piv, "_post_submit")
    timed(piv, "_post_extract")
    timed(piv, "_post_complete")
    timed(piv._plan, "run", "plan.run")
    timed(torch.Tensor, "copy_", "Tensor.copy_")
    timed(torch.Tensor, "to", "Tensor.to")
    timed(torch, "tensor", "torch.tensor")
    timed(piv, "_finish_batch")
    timed(engine, "bmp_unpack")
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
    N = n * reps
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    k = sum(1 for _ in piv.batched(batch))
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f"workers {workers} read threads {threads} depth {depth} batch {batch} {kind}: {N / dt:.0f} pairs/s, {dt * 1e3 / (N / batch):.2f} ms per batch of {batch}, yielded {k}")
    for key in T_:
        print(f"  {key:20s} {T_[key] * 1e3 / (N / batch):7.2f} ms per batch  ({N_[key]} calls)")
    piv.close()
    shutil.rmtree(d, ignore_errors=True)
