"""The resident generator of one hole kind under a kernel trace (development aid):
   rocprofv3 --kernel-trace --stats --output-format csv -d <dir> -- python3 tools/dev/e2e_trace.py clean|runs|spots [pairs] [batch]
prints the wall rate; the trace's per-kernel totals against it say what the GPU was busy with."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torchpiv_amd as T
import e2e_generator as E

if __name__ == "__main__":
    kind = sys.argv[1] if len(sys.argv) > 1 else "clean"
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
    batch = int(sys.argv[3]) if len(sys.argv) > 3 else 64
    A, B = E.make_frames(128, 2048, 2048, kind)
    piv = T.ResidentPIV(A, B, 64, 32, multipass=2, multipass_mode="CWS")
    piv.fill_workers = 8
    order = list(range(128)) * (n // 128)
    sum(1 for _ in piv.batched(batch))
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    k = sum(1 for _ in piv.batched(batch, indices=order))
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f"{kind}: {len(order) / dt:.0f} pairs/s, {dt * 1e3 / (len(order) / batch):.3f} ms per batch of {batch}, yielded {k}, "
          f"launches {len(order) // batch + 128 // batch}")
    piv.close()
