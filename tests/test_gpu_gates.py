"""Are the parity gates real?  The same whole-plan checks that pass on the shipped library must FAIL on a
library whose 32x32 CWS column lerp carries a relative weight error of 1e-3 (tools/diag/libtorchpiv_hip_mutant.so:
xcorr_ws32.hip compiled with -DTPIV_MUTANT_LERP, everything else the shipped objects; built by `make`, loaded
only here, through the development override TPIV_LIB, in ONE child process)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
MUTANT = os.path.join(ROOT, "tools", "diag", "libtorchpiv_hip_mutant.so")


def test_parity_gates_catch_a_perturbed_lerp():
    if not os.path.exists(MUTANT):          # normally built by `make` (build()); a bare checkout builds it here
        subprocess.run(["make", "-C", os.path.join(ROOT, "torchpiv_amd", "csrc"), "-j", "8", "mutant"], check=True,
                       timeout=1200)
    assert os.path.exists(MUTANT), "build the mutant library first (make -C torchpiv_amd/csrc)"
    env = dict(os.environ, TPIV_LIB=MUTANT)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "mutant_probe.py")], env=env, capture_output=True,
                       text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("PROBE ")][-1]
    out = json.loads(line[len("PROBE "):])
    print("  mutant probe:", json.dumps(out, indent=1))
    assert out["lib"] == MUTANT
    # the whole-plan gates (isolation against the oracle fed with the GPU's own fields) notice it at both
    # precisions that use the fast sampling order ...
    assert out["cascade_fast"].startswith("caught"), out
    assert out["cascade_f64"].startswith("caught"), out
    # ... and so does the staging gate of test_fast_staging_close_to_reference_order (<= 1e-4 grey levels)
    assert out["staging_max_abs_diff"] > 1e-4, out
