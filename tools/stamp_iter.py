"""Phase shares of ONE tile-kernel pass (any tile size / mode) from the stamped diagnostic build:
  TPIV_LIB=tools/diag/libtorchpiv_hip_stamps.so python tools/stamp_iter.py CWS 8 4096
runs engine.iterate (pass >= 2 semantics) or engine.pass1 (mode PASS1) on synthetic frames with a
smooth random predictor.  Read the SHARES, not the run time (the stamp fences forbid overlaps)."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torchpiv_amd import engine, synth, _lib

NAMES = ["loop head", "rows wait + convert", "mean + normalise", "fwd row FFT", "transpose 1", "fwd col FFT",
         "cross-spectrum", "inv col FFT", "transpose 2", "inv row FFT", "prefetch issue", "-", "-",
         "peak analysis + store"]
mode = sys.argv[1] if len(sys.argv) > 1 else "CWS"
ws = int(sys.argv[2]) if len(sys.argv) > 2 else 8
size = int(sys.argv[3]) if len(sys.argv) > 3 else 4096
batch = int(sys.argv[4]) if len(sys.argv) > 4 else 4
ov = ws // 2
A, B = synth.make_batch(batch, size, size, device="cuda")
stamps = torch.zeros(32, dtype=torch.int64, device="cuda")
_lib.lib.tpiv_debug_set_stamps.argtypes = [C.c_void_p]
_lib.lib.tpiv_debug_set_stamps(C.c_void_p(stamps.data_ptr()))
nr, nc = engine.field_shape(size, size, ws, ov)
g = torch.Generator(device="cuda").manual_seed(1)
u2 = (torch.rand(batch, nr, nc, device="cuda", dtype=torch.float64, generator=g) - 0.5) * 3.0
v2 = (torch.rand(batch, nr, nc, device="cuda", dtype=torch.float64, generator=g) - 0.5) * 3.0
if mode == "DWS":
    u2, v2 = u2.round(), v2.round()
u0, v0 = 2 * u2, 2 * v2


def run():
    if mode == "PASS1":
        engine.pass1(A, B, ws, ov)
    else:
        engine.iterate(mode, A, B, ws, ov, u0, v0, u2, v2)


run(); torch.cuda.synchronize(); stamps.zero_()
run(); torch.cuda.synchronize()
s = stamps.cpu().numpy().astype(float)
iters, tot = s[16], s[:14].sum()
print(f"{mode} {ws}x{ws} on {size}^2 x{batch}: {int(iters)} wave-iterations, {tot / iters:.0f} cycles per iteration; "
      f"shader clock {s[17] / s[18] * 100:.0f} MHz")
for n, v in zip(NAMES, s[:14]):
    if n != "-":
        print(f"   {n:22s} {v / iters:9.0f} cyc  {100 * v / tot:5.1f} %")
