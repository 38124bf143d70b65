"""Every window size with a compile-time instance of the generic kernel (xcorr_generic_ct_kernel<MODE, N>: in-register
mixed-radix transforms, csrc/fft_mixed.hpp) against the oracle: pass 1 by the golden tests' field rule, and as the second
pass of a 2 N -> N plan in both shift modes by the three gates of cascade_check (reference chain with a capped 16-ulp band,
isolation against the oracle fed with the GPU's own first pass, counted drift).  The goldens cover 24 / 48 (pass 1) and
42 / 28 / 36 (shifted); this file covers the whole instantiated set."""
import numpy as np
import pytest
import torch

from oracle import piv_oracle as O

pytestmark = pytest.mark.gpu

REGISTER_SIZES = [12, 14, 18, 20, 24, 28, 30, 36, 40, 42, 48, 56]       # TPIV_CT_REGISTER_SIZES, csrc/xcorr_generic.hip


@pytest.fixture(scope="module")
def eng():
    assert torch.cuda.is_available(), "GPU tests need a ROCm device"
    from torchpiv_amd import engine
    return engine


def test_the_list_is_the_librarys(eng):
    for n in REGISTER_SIZES:
        plan = eng.Plan(8 * n, 8 * n, 2 * n, n, n_pass=2, mode="CWS", max_batch=1, precision="fast")
        assert plan.kernel_name(1) == f"xcorr_generic_ct_kernel<2, {n}>", (n, plan.kernel_name(1))
        plan.close()
    plan = eng.Plan(352, 352, 88, 44, n_pass=2, mode="DWS", max_batch=1, precision="fast")      # 44 = 4 x 11: no two-factor split
    assert plan.kernel_name(1) == "xcorr_generic_kernel<1, float>", plan.kernel_name(1)
    plan.close()
    plan = eng.Plan(160, 160, 20, 10, n_pass=2, mode="DWS", max_batch=1, precision="fast")      # 10 = 2 x 5: run-time form
    assert plan.geometry[1][0] == 10 and plan.kernel_name(1) == "xcorr_generic_ct_kernel<1, 0>", (plan.geometry, plan.kernel_name(1))
    plan.close()


@pytest.mark.parametrize("n", REGISTER_SIZES)
def test_register_sizes_pass1(eng, n):
    from torchpiv_amd import synth
    from test_gpu_parity import check_fields, dev, near_tie_windows, pass1_constant
    a, b = synth.make_pair(9 * n, 11 * n, 700 + n, kind="vortex", noise=2.0)
    a, b = a.numpy(), b.numpy()
    ov = n // 3
    ru, rv, _, _, rmask = O.pass1(a, b, n, ov, validate=True)
    u, v, inv = eng.pass1(dev(a), dev(b), n, ov, precision="fast")
    e, f = check_fields(u[0], v[0], inv[0], ru, rv, rmask, f"ws{n}", excused=near_tie_windows(a, b, n, ov),
                        constant=pass1_constant(a, b, n, ov))
    print(f"  register-size pass 1, ws {n}: max err {e:.2e} px, mask flips {f}")


@pytest.mark.parametrize("mode", ["CWS", "DWS"])
@pytest.mark.parametrize("n", REGISTER_SIZES)
def test_register_sizes_as_a_shifted_pass(eng, n, mode):
    from torchpiv_amd import synth
    from test_gpu_fullsize import _oracle_fields
    from test_gpu_parity import cascade_check
    a, b = synth.make_pair(8 * n, 10 * n, 800 + n, kind="wavy", noise=2.0)
    geo = [(2 * n, n), (n, n // 2)]
    g = _oracle_fields(a.numpy(), b.numpy(), geo, mode, f"r{n}")
    counts = cascade_check(eng, g, f"r{n}", mode, "fast", geo)
    assert counts[-1][-1] == 15 * 19
