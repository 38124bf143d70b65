"""Resident generator rate against the batch size (development aid)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torchpiv_amd as T
import e2e_generator as E

if __name__ == "__main__":
    n, reps = 128, 64
    for kind in ("spots", "clean"):
        A, B = E.make_frames(n, 2048, 2048, kind)
        for batch, depth in ((128, 2),):
            for workers in (8,):
                piv = T.ResidentPIV(A, B, 64, 32, multipass=2, multipass_mode="CWS")
                piv.fill_workers = workers
                piv.resident_depth = depth
                E.rate(piv.batched(batch), n)
                r, k, rs, _cpu = E.rate3(lambda: piv.batched(batch, indices=list(range(n)) * reps), n * reps)
                print(f"{kind} batch {batch:3d} depth {depth} workers {workers:2d}: {r:8.1f} pairs/s  {rs}  host CPU {_cpu * 1e6:.0f} us/pair", flush=True)
                piv.close()
