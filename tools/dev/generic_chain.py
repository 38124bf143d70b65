"""Timing of a multipass chain with generic window sizes (multipass_scale 1.5: 64/32 -> 42/21 -> 28/14) on 2048^2 frames."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from torchpiv_amd import engine, synth
batch = int(sys.argv[1]) if len(sys.argv) > 1 else 16
scale = float(sys.argv[2]) if len(sys.argv) > 2 else 1.5
mode = sys.argv[3] if len(sys.argv) > 3 else "CWS"
A0, B0 = synth.make_batch(2, 2048, 2048, device="cuda")
A = A0.repeat(batch // 2, 1, 1).contiguous(); B = B0.repeat(batch // 2, 1, 1).contiguous()
for prec in ("exact", "f64", "fast"):
    plan = engine.Plan(2048, 2048, 64, 32, n_pass=3, mode=mode, pass_scale=scale, max_batch=batch, precision=prec)
    out = plan.run(A, B); torch.cuda.synchronize()
    plan.set_timing(True)
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
    ev[0].record()
    for i in range(3):
        plan.run(A, B, out=out); ev[i + 1].record()
    torch.cuda.synchronize()
    ts = sorted(ev[i].elapsed_time(ev[i + 1]) for i in range(3))
    tm, n = plan.get_timing()
    print(f"{mode} x{scale} {prec}: geometry {plan.geometry}, {ts[1]:.2f} ms per {batch} pairs = {batch / ts[1] * 1000:.0f} pairs/s; per-kernel us/pair",
          {k: round(v / batch * 1000, 1) for k, v in tm.items()}, [plan.kernel_name(p) for p in range(3)])
    plan.close()
