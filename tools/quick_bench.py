"""Quick timing of the multipass plan on synthetic frames (development aid; bench.py is the contract)."""
import argparse, sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torchpiv_amd import engine, synth

ap = argparse.ArgumentParser()
ap.add_argument("--size", type=int, default=2048)
ap.add_argument("--ws", type=int, default=64)
ap.add_argument("--passes", type=int, default=2)
ap.add_argument("--mode", default="CWS")
ap.add_argument("--batch", type=int, default=16)
ap.add_argument("--distinct", type=int, default=2)
ap.add_argument("--iters", type=int, default=5)
ap.add_argument("--kind", default="wavy")
ap.add_argument("--noise", type=float, default=0.0)
ap.add_argument("--width", type=int, default=0, help="frame width when it differs from --size (row-pitch experiments)")
ap.add_argument("--precision", default="fast", choices=("fast", "f64", "reference", "exact"))
ap.add_argument("--scale", type=float, default=2.0, help="multipass_scale (window size of pass p+1 = int(ws_p // scale))")
a = ap.parse_args()
H = a.size
W = a.width or a.size
A0, B0 = synth.make_batch(a.distinct, H, W, device="cuda", kind=a.kind, noise=a.noise)
A = A0.repeat((a.batch + a.distinct - 1) // a.distinct, 1, 1)[:a.batch].contiguous()
B = B0.repeat((a.batch + a.distinct - 1) // a.distinct, 1, 1)[:a.batch].contiguous()
plan = engine.Plan(H, W, a.ws, a.ws // 2, n_pass=a.passes, mode=a.mode, pass_scale=a.scale, max_batch=a.batch, precision=a.precision)
out = plan.run(A, B)
torch.cuda.synchronize()
for p in range(a.passes):
    print("pass", p, plan.geometry[p])
ev = [torch.cuda.Event(enable_timing=True) for _ in range(a.iters + 1)]
ev[0].record()
for i in range(a.iters):
    plan.run(A, B, out=out)
    ev[i + 1].record()
torch.cuda.synchronize()
ts = [ev[i].elapsed_time(ev[i + 1]) for i in range(a.iters)]
t = sorted(ts)[len(ts) // 2]
print(f"size {H} ws {a.ws} passes {a.passes} {a.mode} batch {a.batch} precision {a.precision}: {t:.3f} ms/batch, "
      f"{t / a.batch * 1000:.1f} us/pair, {a.batch / t * 1000:.1f} pairs/s  (all: {[round(x, 2) for x in ts]})")
plan.set_timing(True)
for i in range(3):
    plan.run(A, B, out=out)
torch.cuda.synchronize()
tm, n = plan.get_timing()
print("per-kernel ms per batch:", {k: round(v, 3) for k, v in tm.items()}, " us/pair:",
      {k: round(v / a.batch * 1000, 1) for k, v in tm.items()})
u, v, inv = out
if plan.exact_capable():
    print("exact pass 1, ms per batch:", {k: round(v, 3) for k, v in plan.exact_timing().items()})
    n_fb = plan.exact_fallbacks()
    n_w = a.batch * plan.geometry[0][2] * plan.geometry[0][3]
    print(f"exact: {n_fb} of {n_w} first-pass windows took the float64 transform ({100.0 * n_fb / n_w:.3f} %)")
print("invalid frac", inv.float().mean().item(), "u mean", u.mean().item(), "v mean", v.mean().item())
