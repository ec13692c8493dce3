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
