"""Child process of tests/test_gpu_gates.py: runs the whole-plan parity gates on whatever library TPIV_LIB
names (there: tools/diag/libtorchpiv_hip_mutant.so, whose 32x32 CWS column lerp carries a 1e-3 weight error)
and prints what each gate said.  Not a test module itself."""
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(HERE))

import numpy as np  # noqa: E402
import torch  # noqa: E402


def main():
    import test_gpu_parity as P
    from oracle import piv_oracle as O
    from test_gpu_fullsize import _oracle_fields
    from torchpiv_amd import _lib, engine, synth
    out = {"lib": _lib.LIB_PATH}
    a, b = synth.make_pair(1024, 1024, 321, kind="wavy", noise=3.0)
    a, b = a.numpy(), b.numpy()
    geo = [(64, 32), (32, 16)]
    g = _oracle_fields(a, b, geo, "CWS", "probe")
    for precision in ("fast", "f64"):
        try:
            P.cascade_check(engine, g, "probe", "CWS", precision, geo)
            out["cascade_" + precision] = "passed"
        except AssertionError as exc:
            out["cascade_" + precision] = "caught: " + str(exc)[:300]
    # the staging gate (test_fast_staging_close_to_reference_order)
    rng = np.random.default_rng(5)
    H, W, ws, ov = 168, 196, 32, 16
    fa = rng.integers(0, 256, size=(H, W)).astype(np.uint8)
    nr, nc = O.field_shape((H, W), ws, ov)
    vx = torch.from_numpy(rng.uniform(-9, 9, nr * nc)).cuda().view(1, nr, nc)
    vy = torch.from_numpy(rng.uniform(-9, 9, nr * nc)).cuda().view(1, nr, nc)
    _, _, _, w_ref, _ = engine.debug_pass("CWS", P.dev(fa), P.dev(fa), ws, ov, vx, vy, precision="reference")
    _, _, _, w_fast, _ = engine.debug_pass("CWS", P.dev(fa), P.dev(fa), ws, ov, vx, vy, precision="fast")
    out["staging_max_abs_diff"] = float((w_ref - w_fast).abs().max().item())
    print("PROBE " + json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
