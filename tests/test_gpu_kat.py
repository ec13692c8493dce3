"""The known-answer cases of tests/kat_cases.py asserted on the HIP path (librtr_hip.so through its C ABI): closed-form
expectations derived from the reference's source lines, with NO oracle in between -- the reference holds no fixtures and
cannot be built here, so these are the only oracle-independent pins the product has."""
import pytest

import kat_cases

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def hip():
    b = kat_cases.HipBackend(0)  # fails loudly without a device: there is no fallback
    yield b
    b.close()


@pytest.mark.parametrize("case", kat_cases.FRAME_CASES + kat_cases.RAY_CASES + kat_cases.HOST_CASES, ids=lambda c: c.__name__)
def test_known_answer_on_the_hip_path(hip, case):
    case(hip)


def test_the_oracle_is_not_loaded_by_the_product_path():
    """The cases above must not have pulled the checker in: nothing under raytracingrust_amd/ imports oracle/."""
    import sys

    import raytracingrust_amd  # noqa: F401

    mods = [m for m in sys.modules if m.startswith("raytracingrust_amd")]
    for m in mods:
        src = getattr(sys.modules[m], "__file__", None)
        if src and src.endswith(".py"):
            text = open(src).read()
            assert "import pyoracle" not in text and "librt_oracle" not in text, m
