"""Native batch reader (tpiv_read_files) alone and against a concurrent H2D stream.  (development aid)"""
import os, sys, time, tempfile, threading, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from torchpiv_amd._lib import lib

print("cpu_count", os.cpu_count(), "affinity", len(os.sched_getaffinity(0)))
for f in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpuset.cpus.effective", "/proc/loadavg"):
    try:
        print(f, open(f).read().strip())
    except OSError as e:
        print(f, e)
os.system("lscpu | grep -i -E 'numa|model name|socket' ; free -g | head -2")
n, size = 64, 4 * 1024 * 1024 + 1078
d = tempfile.mkdtemp()
rng = np.random.default_rng(0)
paths = []
for i in range(4 * n):
    p = os.path.join(d, f"f{i:04d}.bin")
    open(p, "wb").write(rng.integers(0, 256, size, dtype=np.uint8).tobytes())
    paths.append(p)
cap = (size + 4095) // 4096 * 4096
bufs = [torch.empty(n, cap, dtype=torch.uint8).pin_memory() for _ in range(3)]
sizes = np.empty(n, dtype=np.int64)


def read(k, threads):
    ps = paths[(k % 4) * n:(k % 4 + 1) * n]
    arr = (C.c_char_p * n)(*[os.fsencode(p) for p in ps])
    lib.tpiv_read_files(arr, n, C.c_void_p(bufs[k % 3].data_ptr()), cap, threads, sizes.ctypes.data_as(C.POINTER(C.c_longlong)))


for th in (2, 4, 8, 12, 16, 24, 32):
    read(0, th)
    t = time.perf_counter()
    for k in range(8):
        read(k, th)
    dt = (time.perf_counter() - t) / 8
    print(f"read alone, {th:2d} threads: {dt * 1e3:6.2f} ms per 64 files  {n * size / dt / 1e9:6.1f} GB/s")
g = [torch.empty(n, cap, dtype=torch.uint8, device="cuda") for _ in range(2)]
torch.cuda.synchronize()
t = time.perf_counter()
for k in range(8):
    g[k % 2].copy_(bufs[k % 3], non_blocking=True)
torch.cuda.synchronize()
dt = (time.perf_counter() - t) / 8
print(f"H2D alone: {dt * 1e3:.2f} ms per 64 slots  {n * cap / dt / 1e9:.1f} GB/s")
for th in (4, 8, 12, 16):
    stop = False

    def pump():
        k = 0
        while not stop:
            g[k % 2].copy_(bufs[2], non_blocking=True)
            torch.cuda.synchronize()
            k += 1
        pump.k = k
    tp = threading.Thread(target=pump)
    t0 = time.perf_counter()
    tp.start()
    t = time.perf_counter()
    for k in range(8):
        read(k * 3, th)         # buffers 0 only
    dt = (time.perf_counter() - t) / 8
    stop = True
    tp.join()
    dtp = (time.perf_counter() - t0) / pump.k
    print(f"read {th:2d} threads with H2D running: read {dt * 1e3:6.2f} ms ({n * size / dt / 1e9:5.1f} GB/s), H2D {dtp * 1e3:.2f} ms ({n * cap / dtp / 1e9:.1f} GB/s)")
import shutil
shutil.rmtree(d)
