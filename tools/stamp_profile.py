"""Per-phase cycle shares of the tile kernels from the stamped diagnostic build
(make -C torchpiv_amd/csrc stamps; run with TPIV_LIB=tools/diag/libtorchpiv_hip_stamps.so).
Read the SHARES, not the run time: the stamp fences forbid overlaps the real kernel has."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torchpiv_amd import engine, synth, _lib

NAMES = ["loop head", "rows wait + convert", "mean + normalise", "fwd row FFT", "transpose 1", "fwd col FFT",
         "cross-spectrum", "inv col FFT", "transpose 2", "inv row FFT", "prefetch issue", "min/map/peak1",
         "second peak", "subpixel+store"]
H = W = 2048
A, B = synth.make_batch(4, H, W, device="cuda")
A = A.repeat(8, 1, 1).contiguous(); B = B.repeat(8, 1, 1).contiguous()
stamps = torch.zeros(32, dtype=torch.int64, device="cuda")
_lib.lib.tpiv_debug_set_stamps.argtypes = [C.c_void_p]
_lib.lib.tpiv_debug_set_stamps(C.c_void_p(stamps.data_ptr()))
mode = sys.argv[1] if len(sys.argv) > 1 else "CWS"
plan = engine.Plan(H, W, 64, 32, n_pass=2, mode=mode, max_batch=32)
u0 = None
for label, run in (("pass 1 (64x64)", "p1"), (f"pass 2 (32x32 {mode})", "p2")):
    plan.run(A, B); torch.cuda.synchronize()
    stamps.zero_()
    if run == "p1":
        engine.pass1(A, B, 64, 32)
    else:
        # rerun the whole plan but subtract pass-1 stamps measured separately
        s1 = stamps.clone()
        engine.pass1(A, B, 64, 32); torch.cuda.synchronize(); s1 = stamps.clone(); stamps.zero_()
        plan.run(A, B); torch.cuda.synchronize(); stamps -= s1
    torch.cuda.synchronize()
    s = stamps.cpu().numpy().astype(float)
    iters = s[16]
    tot = s[:14].sum()
    print(f"{label}: {int(iters)} wave-iterations, {tot / iters:.0f} cycles per iteration; "
          f"shader clock {s[17] / s[18] * 100:.0f} MHz (s_memtime / s_memrealtime)")
    for n, v in zip(NAMES, s[:14]):
        print(f"   {n:22s} {v / iters:9.0f} cyc  {100 * v / tot:5.1f} %")
