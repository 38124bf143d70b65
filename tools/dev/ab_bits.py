"""Do two builds of the library produce the same BITS?  (development aid for changes that must not move a result)
   python tools/dev/ab_bits.py dump out.npz     (with TPIV_LIB=<build>)      -> fields of a fixed set of plans on seeded frames
   python tools/dev/ab_bits.py cmp a.npz b.npz                               -> per-plan count of differing words"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np

CASES = [  # (H, W, ws, passes, mode, precision, kind, noise)
    (512, 512, 64, 2, "CWS", "exact", "wavy", 2.0), (512, 512, 64, 2, "DWS", "fast", "wavy", 2.0),
    (512, 512, 32, 3, "CWS", "fast", "wavy", 4.0), (512, 512, 32, 3, "DWS", "exact", "wavy", 4.0),
    (512, 768, 128, 2, "CWS", "fast", "wavy", 2.0), (384, 512, 16, 2, "CWS", "fast", "wavy", 6.0),
    (512, 512, 64, 1, "CWS", "fast", "wavy", 30.0), (256, 256, 32, 1, "CWS", "fast", "noise", 0.0),
    (512, 512, 48, 2, "CWS", "fast", "wavy", 2.0),
]

if sys.argv[1] == "dump":
    import torch
    from torchpiv_amd import engine, synth
    out = {}
    for i, (H, W, ws, n_pass, mode, prec, kind, noise) in enumerate(CASES):
        try:
            A, B = synth.make_batch(3, H, W, device="cuda", kind=kind, noise=noise)
        except Exception:                                           # noqa: BLE001 -- kinds this synth does not know
            g = torch.Generator(device="cuda").manual_seed(7 + i)
            A = torch.randint(0, 256, (3, H, W), dtype=torch.uint8, device="cuda", generator=g)
            B = torch.randint(0, 256, (3, H, W), dtype=torch.uint8, device="cuda", generator=g)
        plan = engine.Plan(H, W, ws, ws // 2, n_pass=n_pass, mode=mode, max_batch=3, precision=prec)
        u, v, inv = plan.run(A, B)
        torch.cuda.synchronize()
        out[f"u{i}"], out[f"v{i}"], out[f"m{i}"] = u.cpu().numpy(), v.cpu().numpy(), inv.cpu().numpy()
    np.savez(sys.argv[2], **out)
    print("dumped", len(CASES), "plans to", sys.argv[2], "with", os.environ.get("TPIV_LIB", "the in-tree library"))
else:
    a, b = np.load(sys.argv[2]), np.load(sys.argv[3])
    bad = 0
    for i, c in enumerate(CASES):
        d = [int((a[k + str(i)].view(np.uint8) != b[k + str(i)].view(np.uint8)).reshape(-1).sum()) for k in "uvm"]
        n = a["u" + str(i)].size
        bad += sum(d)
        print(c, "differing bytes u/v/mask:", d, "of", n, "vectors", "nan:", int(np.isnan(a["u" + str(i)]).sum()))
    print("IDENTICAL" if bad == 0 else "DIFFERENT")
    sys.exit(1 if bad else 0)
