"""Per-kernel times of the 64/32 two-pass CWS plan on the generator's frame kinds (development aid): do dead spots cost the passes anything?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torchpiv_amd import engine
import e2e_generator as E

if __name__ == "__main__":
    for kind in ("clean", "spots", "runs"):
        A, B = E.make_frames(64, 2048, 2048, kind)
        plan = engine.Plan(2048, 2048, 64, 32, n_pass=2, mode="CWS", max_batch=64, precision="exact")
        out = plan.run(A, B)
        plan.set_timing(True)
        for _ in range(5):
            plan.run(A, B, out=out)
        torch.cuda.synchronize()
        tm, n = plan.get_timing()
        print(kind, {k: round(v, 3) for k, v in tm.items()}, "invalid", float(out[2].float().mean()))
        plan.close()
