"""File path rates measured repeatedly in one process (is the first pass over the files slower?).  (development aid)"""
import os, sys, time, tempfile, shutil
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import torchpiv_amd as T
import e2e_generator as E

if __name__ == "__main__":
    from PIL import Image
    n, reps = 128, 8
    A, B = E.make_frames(n, 2048, 2048, "spots")
    d = tempfile.mkdtemp()
    for i in range(n):
        Image.fromarray(A[i].cpu().numpy(), "L").save(os.path.join(d, f"img{i:05d}_a.bmp"))
        Image.fromarray(B[i].cpu().numpy(), "L").save(os.path.join(d, f"img{i:05d}_b.bmp"))
        for r_ in range(1, reps):
            for s_ in "ab":
                os.link(os.path.join(d, f"img{i:05d}_{s_}.bmp"), os.path.join(d, f"img{i + r_ * n:05d}_{s_}.bmp"))
    del A, B
    fp = T.OfflinePIV(d, "cuda:0", "bmp", 64, 32, multipass=2, multipass_mode="CWS")
    fp.fill_workers = 8
    for batch in (32, 64):
        fp.call_batch = batch
        E.rate(fp.batched(batch, indices=range(n)), n)
        for rnd in range(3):
            r, k = E.rate(fp.batched(batch), n * reps)
            print(f"round {rnd} batched({batch}): {r:8.1f} pairs/s, yielded {k}", flush=True)
            r, k = E.rate(fp(), n * reps)
            print(f"round {rnd} __call__ ({batch}): {r:8.1f} pairs/s, yielded {k}", flush=True)
    fp.close()
    shutil.rmtree(d, ignore_errors=True)
