"""How fast do 4 MB files come out of the page cache into (pinned) host memory, and up to the GPU?  (development aid)"""
import os, sys, time, tempfile, threading
from concurrent.futures import ThreadPoolExecutor
import numpy as np, torch

n, size = 128, 4 * 1024 * 1024 + 1078
d = tempfile.mkdtemp()
blob = np.random.default_rng(0).integers(0, 256, size, dtype=np.uint8).tobytes()
paths = []
for i in range(n):
    p = os.path.join(d, f"f{i:04d}.bin")
    open(p, "wb").write(blob)
    paths.append(p)
print("cpu_count", os.cpu_count(), "affinity", len(os.sched_getaffinity(0)))
cap = (size + 4095) // 4096 * 4096
pinned = torch.empty(n, cap, dtype=torch.uint8).pin_memory()
plain = torch.empty(n, cap, dtype=torch.uint8)


def rd(args):
    p, buf = args
    with open(p, "rb", buffering=0) as f:
        f.readinto(memoryview(buf)[:size])


for name, dst in (("pinned", pinned), ("pageable", plain)):
    arr = dst.numpy()
    for th in (1, 4, 8, 16, 32):
        with ThreadPoolExecutor(th) as ex:
            list(ex.map(rd, [(paths[i], arr[i]) for i in range(n)]))      # warm
            t = time.perf_counter()
            list(ex.map(rd, [(paths[i], arr[i]) for i in range(n)]))
            dt = time.perf_counter() - t
        print(f"{name:9s} threads {th:2d}: {n * size / dt / 1e9:6.2f} GB/s  ({n / 2 / dt:7.0f} pairs/s)")
if torch.cuda.is_available():
    g = torch.empty(n, cap, dtype=torch.uint8, device="cuda")
    for _ in range(2):
        torch.cuda.synchronize(); t = time.perf_counter()
        g.copy_(pinned, non_blocking=True); torch.cuda.synchronize()
        dt = time.perf_counter() - t
    print(f"H2D pinned: {n * cap / dt / 1e9:.1f} GB/s")
