"""Per-phase cycle shares of the float64 first-pass kernel for 64x64 windows (xcorr_f64.hip) from
the stamped diagnostic build (make -C torchpiv_amd/csrc stamps; TPIV_LIB=tools/diag/libtorchpiv_hip_stamps.so).
Read the SHARES: the stamp fences forbid overlaps the real kernel has, and the counters are per wavefront (both
wavefronts of a window add theirs)."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torchpiv_amd import engine, synth, _lib

NAMES = ["sums + rows forward", "transposition 1", "columns forward", "cross-spectrum", "columns inverse",
         "transposition 2", "rows inverse", "peak + prefetch issue"]
H = W = 2048
A, B = synth.make_batch(4, H, W, device="cuda")
A = A.repeat(8, 1, 1).contiguous(); B = B.repeat(8, 1, 1).contiguous()
stamps = torch.zeros(32, dtype=torch.int64, device="cuda")
_lib.lib.tpiv_debug_set_stamps.argtypes = [C.c_void_p]
_lib.lib.tpiv_debug_set_stamps(C.c_void_p(stamps.data_ptr()))
engine.pass1(A, B, 64, 32, precision="f64"); torch.cuda.synchronize()
stamps.zero_()
engine.pass1(A, B, 64, 32, precision="f64"); torch.cuda.synchronize()
s = stamps.cpu().numpy().astype(float)
iters, tot = s[16], s[:8].sum()
print(f"float64 pass 1 (64x64): {int(iters)} wave-iterations, {tot / iters:.0f} cycles per iteration; "
      f"shader clock {s[17] / s[18] * 100:.0f} MHz (s_memtime / s_memrealtime)")
# (third generation: the inverse column transform is scheduled behind its stamp, into the phase of the T2 write -- the two
#  are reported together)
vals = list(s[:8])
vals[5] += vals[4]
names = list(NAMES)
names[5] = "columns inverse + transposition 2"
for k, (n, v) in enumerate(zip(names, vals)):
    if k == 4:
        continue
    print(f"   {n:34s} {v / iters:9.0f} cyc  {100 * v / tot:5.1f} %")
