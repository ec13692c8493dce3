"""Device self-tests of arithmetic building blocks the kernels substitute for compiler expansions."""
import pytest

pytestmark = pytest.mark.gpu


def test_rt_sqrt_equals_sqrtf_for_every_f32(gpu_renderer):
    """rt_sqrt (v_sqrt_f32 + neighbour test without the denormal/class wrapping) must equal the compiler's
    correctly rounded sqrtf() for all 2^32 bit patterns -- bit for bit, NaN for NaN."""
    assert gpu_renderer.selftest_sqrt() == 0
