"""End-to-end rate of the drop-in generator on BMP files (decode + H2D + kernels + D2H + host hole fill)."""
import os, sys, tempfile, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from PIL import Image
import torchpiv_amd as T
from torchpiv_amd import synth

n = int(sys.argv[1]) if len(sys.argv) > 1 else 16
H = W = 2048
d = tempfile.mkdtemp()
for i in range(n):
    a, b = synth.make_pair(H, W, i, noise=3.0, device="cuda")
    a, b = a.cpu().numpy().copy(), b.cpu().numpy().copy()
    a[100:140, 200:260] = 0; b[100:140, 200:260] = 0          # a dead patch -> some invalid vectors
    Image.fromarray(a, "L").save(os.path.join(d, f"img{i:04d}_a.bmp"))
    Image.fromarray(b, "L").save(os.path.join(d, f"img{i:04d}_b.bmp"))
piv = T.OfflinePIV(d, "cuda:0", "bmp", 64, 32, multipass=2, multipass_mode="CWS")
t0 = time.perf_counter(); r = list(piv()); t1 = time.perf_counter()
print(f"generator: {len(r)}/{len(piv)} pairs in {t1 - t0:.2f} s -> {len(piv) / (t1 - t0):.1f} pairs/s (first call incl. plan creation)")
t0 = time.perf_counter(); r = list(piv()); t1 = time.perf_counter()
print(f"generator: {len(r)}/{len(piv)} pairs in {t1 - t0:.2f} s -> {len(piv) / (t1 - t0):.1f} pairs/s")
t0 = time.perf_counter(); r = list(piv.batched(8)); t1 = time.perf_counter()
print(f"batched(8): {len(r)}/{len(piv)} pairs in {t1 - t0:.2f} s -> {len(piv) / (t1 - t0):.1f} pairs/s")
# where does the time go (one pair)
import cProfile, pstats
pr = cProfile.Profile(); pr.enable(); list(piv()); pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(14)
