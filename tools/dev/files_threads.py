"""File path rate against reader threads / fill workers (development aid)."""
import os, sys, tempfile, shutil
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
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
    for threads, workers in ((8, 8), (12, 8), (12, 6), (16, 4), (6, 8), (8, 8)):
        fp = T.OfflinePIV(d, "cuda:0", "bmp", 64, 32, multipass=2, multipass_mode="CWS")
        fp.fill_workers, fp.read_threads = workers, threads
        E.rate(fp.batched(32, indices=range(n)), n)
        r, k, rs = E.rate3(lambda: fp.batched(32), n * reps, trials=4)
        print(f"read threads {threads:2d} fill workers {workers}: {r:8.1f} pairs/s  {rs}", flush=True)
        fp.close()
    shutil.rmtree(d, ignore_errors=True)
