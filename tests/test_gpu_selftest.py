"""Device self-tests of arithmetic building blocks the kernels substitute for compiler expansions."""
import pytest

pytestmark = pytest.mark.gpu


def test_rt_sqrt_equals_sqrtf_for_every_f32(gpu_renderer):
    """rt_sqrt (v_sqrt_f32 + neighbour test without the denormal/class wrapping) must equal the compiler's
    correctly rounded sqrtf() for all 2^32 bit patterns -- bit for bit, NaN for NaN."""
    assert gpu_renderer.selftest_sqrt() == 0


def test_torch_still_finds_the_gpu_when_the_library_is_loaded_first():
    """PyTorch-ROCm bundles its own HIP runtime; when librtr_hip.so (linked against /opt/rocm) initialised the GPU
    first, torch.cuda reported no device.  _capi.load() imports torch ahead of the library; this runs the bad order
    in a fresh interpreter: renderer first, torch after."""
    import os
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = (
        "import sys; sys.path.insert(0, %r); sys.path.insert(0, %r)\n"
        "import raytracingrust_amd as rt\n"
        "from helpers import benchmark_scene\n"
        "r = rt.Renderer(0); r.upload_scene(benchmark_scene(1)); r.render(16, 16, seed=1)\n"
        "import torch\n"
        "assert torch.cuda.is_available()\n"
        "print(int(torch.ones(4, device='cuda').sum().item()))\n" % (root, os.path.join(root, "tests")))
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and out.stdout.strip().endswith("4"), (out.stdout[-500:], out.stderr[-2000:])


@pytest.mark.parametrize("denominator", [1919.0, 1079.0, 3839.0, 2159.0, 1.0, 7.0, 99.0, 255.0, 2.0 ** 31,
                                         0.3, 0.5, 100.0, 10.0, 0.45, 1.0e-6, 3.0e11, -0.5, 2.0 ** -20, 2.0 ** 40])
def test_shared_denominator_division_is_the_correctly_rounded_quotient(gpu_renderer, denominator):
    """The camera's (x + xi)/(W-1), (y + xi)/(H-1) and the sphere normal's (p - c)/r reuse one prepared reciprocal
    (rtr_kernels.hip shared_div).  Swept on the device against `/` over every f32 numerator in [2^-40, 2^60], both
    signs, plus +0: not one differing bit."""
    r = gpu_renderer
    assert r.selftest_div(denominator, 2.0 ** -40, 2.0 ** 60) == 0
    assert r.selftest_div(denominator, -(2.0 ** -40), -(2.0 ** 60)) == 0  # negative floats: bit patterns ascend with |n|
    assert r.selftest_div(denominator, 0.0, 0.0) == 0  # +0 (the camera's x + xi at pixel 0); -0 is outside the contract


def test_shared_denominator_division_refuses_denominators_outside_its_range(gpu_renderer):
    for bad in (0.0, 2.0 ** -21, 2.0 ** 41, float("inf"), float("nan")):
        assert gpu_renderer.selftest_div(bad, 1.0, 1.0) >= 1  # reported as a mismatch: the kernels divide plainly there
