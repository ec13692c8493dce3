import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.dirname(os.path.abspath(__file__))):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session", autouse=True)
def _built():
    """Build the HIP library and the oracle once per session (no-ops when up to date)."""
    from raytracingrust_amd import build as rtr_build
    import pyoracle

    rtr_build.build()
    pyoracle.build()


@pytest.fixture(scope="session")
def gpu_renderer():
    import raytracingrust_amd as rt

    r = rt.Renderer(0)  # fails loudly without a device
    yield r
    r.close()
