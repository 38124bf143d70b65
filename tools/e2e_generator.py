"""End-to-end rate of the drop-in generator (development aid; `bench.py --e2e` prints the same figures as JSON).

  resident   ResidentPIV.batched over frames already in HBM: passes + device post-validation + counted
             host fallbacks + flip/scale + yield
  files      OfflinePIV.batched over 8-bit BMP files: read into pinned staging, upload of the raw bytes,
             device unpack, then as above
"""
import os
import sys
import tempfile
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import torchpiv_amd as T
from torchpiv_amd import synth


def make_frames(n, H, W, kind):
    """kind: 'clean' (no invalid vector: every pair is dropped by the reference's quirk), 'runs' (short
    straight runs of dead windows: device-complete post-validation), 'spots' (isolated dead spots: the
    co-circular case that needs the host triangulation)."""
    A, B = synth.make_batch(n, H, W, device="cuda", noise=2.0)
    g = torch.Generator(device="cpu").manual_seed(7)
    for i in range(n):
        if kind == "clean":
            continue
        for _ in range(6):
            y = int(torch.randint(100, H - 200, (1,), generator=g))
            x = int(torch.randint(100, W - 200, (1,), generator=g))
            if kind == "runs":            # a 40 x 72 px black bar: two horizontally adjacent dead 32x32 windows
                A[i, y:y + 40, x:x + 72] = 0
                B[i, y:y + 40, x:x + 72] = 0
            else:                          # a 40 x 40 px black spot: one dead 32x32 window at 16 px pitch
                A[i, y:y + 40, x:x + 40] = 0
                B[i, y:y + 40, x:x + 40] = 0
    return A, B


def rate(gen, n):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    k = sum(1 for _ in gen)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    return n / dt, k


def rate3(make_gen, n, trials=3):
    """Median of `trials` runs behind one untimed run (the first pass over a case pays for page faults, the plan's first
    launches and the worker pool's start: round 3 counted it as a trial and the medians hid a +-20 % spread); returns
    (median rate, yielded, [all rates], host CPU seconds per pair: user + system time of this process, its threads and
    its fill-worker processes over the timed trials)."""
    from torchpiv_amd import hostcfg
    rate(make_gen(), n)
    rs = []
    c0 = hostcfg.tree_cpu_seconds()
    for _ in range(trials):
        r, k = rate(make_gen(), n)
        rs.append(r)
    cpu = (hostcfg.tree_cpu_seconds() - c0) / (trials * n)
    return sorted(rs)[len(rs) // 2], k, [round(r_) for r_ in rs], cpu


def main(n=128, H=2048, W=2048, batch=32, workers=0, read_threads=0, files=True, budget_s=None, precision="exact", reps=8,
         resident_batch=128, resident_reps=64):
    """Rates in pairs/s per case; `files` adds the BMP cases (skipped once `budget_s` seconds are spent).  Every case
    streams n * reps pairs (the n distinct pairs `reps` times over: 128 pairs alone last some 20 ms, which measures the
    pipeline's fill and drain, not its rate); the file cases hard-link the n pairs' files under n * reps names.  The resident
    cases stream n * resident_reps pairs: the host side is a three-stage pipeline (launch, census + triangulations handed to the
    workers, patch + hand-out), and with 64 pairs per launch 1024 pairs are 16 launches, two of them fill and drain -- round 5
    measured 13.6 k pairs/s over 1024 pairs and 15.3 k over 2048 with the same code at 64 pairs per launch."""
    # (resident frames: 128 pairs per launch -- round 5, 65 536 pairs streamed: isolated spots 15.6 k pairs/s at 64, 16.2 k at 128
    #  (the passes themselves: 16.0 k / 16.6 k / 16.9 k pairs/s at 64 / 128 / 256 pairs per launch); the file path is bound by the
    #  PCIe link at any batch size and keeps 32, the staging buffers' size)
    out = {"precision": precision, "pairs_streamed": n * reps, "resident_pairs_streamed": n * resident_reps,
           "resident_batch": resident_batch, "files_batch": batch}
    order = list(range(n)) * resident_reps
    t_start = time.perf_counter()
    for kind in ("clean", "runs", "spots"):
        A, B = make_frames(n, H, W, kind)
        piv = T.ResidentPIV(A, B, 64, 32, multipass=2, multipass_mode="CWS", precision=precision)
        piv.fill_workers = workers
        rate(piv.batched(resident_batch), n)             # warm-up: plan creation
        piv.reset_stats()
        r, k, rs, cpu = rate3(lambda: piv.batched(resident_batch, indices=order), n * resident_reps)
        out[kind] = r
        out.setdefault("trials", {})[kind] = rs
        out.setdefault("host_cpu_s_per_pair", {})[kind] = cpu
        st_ = {k_: v_ // 4 for k_, v_ in piv.stats.items()}
        out.setdefault("stats", {})[kind] = dict(st_, yielded=k)
        print(f"resident {kind:6s}: {r:8.1f} pairs/s, median of {rs} ({k} of {n * resident_reps} yielded), host CPU {cpu * 1e6:.0f} us/pair  {st_}")
        over = budget_s is not None and time.perf_counter() - t_start > budget_s
        if kind == "spots" and files and not over:
            from PIL import Image
            d = tempfile.mkdtemp()
            for i in range(n):
                Image.fromarray(A[i].cpu().numpy(), "L").save(os.path.join(d, f"img{i:05d}_a.bmp"))
                Image.fromarray(B[i].cpu().numpy(), "L").save(os.path.join(d, f"img{i:05d}_b.bmp"))
                for r_ in range(1, reps):
                    for s_ in "ab":
                        os.link(os.path.join(d, f"img{i:05d}_{s_}.bmp"), os.path.join(d, f"img{i + r_ * n:05d}_{s_}.bmp"))
            fp = T.OfflinePIV(d, "cuda:0", "bmp", 64, 32, multipass=2, multipass_mode="CWS", precision=precision)
            fp.fill_workers = workers
            if read_threads:
                fp.read_threads = read_threads
            rate(fp.batched(batch, indices=range(n)), n)
            fp.reset_stats()
            r, k, rs, cpu = rate3(lambda: fp.batched(batch), n * reps)
            out["files"] = r
            out["trials"]["files"] = rs
            out["host_cpu_s_per_pair"]["files"] = cpu
            print(f"files    {kind:6s}: {r:8.1f} pairs/s, median of {rs} ({k} of {n * reps} yielded; 8-bit BMP in the page cache), host CPU {cpu * 1e6:.0f} us/pair")
            r, k, rs, cpu = rate3(lambda: fp(), n * reps)
            out["files_call"] = r
            out["trials"]["files_call"] = rs
            out["host_cpu_s_per_pair"]["files_call"] = cpu
            print(f"files    __call__ (the reference's generator API; reads ahead {fp.call_batch} pairs per launch): {r:8.1f} pairs/s, median of {rs} ({k} yielded)")
            if budget_s is None:
                fp.call_batch = 1
                r, k = rate(fp(), n * reps)
                print(f"files    __call__ with call_batch = 1 (one pair per launch, host decode): {r:8.1f} pairs/s")
            fp.close()
            import shutil
            shutil.rmtree(d, ignore_errors=True)
        piv.close()
        del A, B
    return out


if __name__ == "__main__":
    main(int(sys.argv[1]) if len(sys.argv) > 1 else 128, workers=int(sys.argv[2]) if len(sys.argv) > 2 else 0,
         read_threads=int(sys.argv[3]) if len(sys.argv) > 3 else 0,
         precision=sys.argv[4] if len(sys.argv) > 4 else "exact")
