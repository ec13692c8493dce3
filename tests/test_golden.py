"""Golden regression vectors (tests/golden/*.npz, written by tests/golden/make_golden.py with the oracle).
CPU: the oracle still reproduces them (guards the checker).  GPU: the HIP path reproduces them bit for bit
on sphere scenes and within tolerance on mesh scenes."""
import os

import numpy as np
import pytest

import pyoracle
import raytracingrust_amd as rt
from golden_cases import CASES
from helpers import compare

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load(name):
    return np.load(os.path.join(GOLDEN, name + ".npz"))


@pytest.mark.parametrize("name", sorted(CASES))
def test_oracle_reproduces_golden(name):
    builder, w, h, seed = CASES[name]
    g = load(name)
    scene = builder()
    if "sky" in g:
        assert np.array_equal(scene.background.image, g["sky"])  # the synthetic sky is host-independent
    rgba, mean, st = pyoracle.OracleScene(scene, trig_mode=pyoracle.TRIG_POLY).render(w, h, seed, threads=0)
    assert st.segments == int(g["segments"]) and st.tri_tests == int(g["tri_tests"]) and st.box_tests == int(g["box_tests"])
    assert np.array_equal(mean.view(np.uint32), g["mean"].view(np.uint32))
    assert np.array_equal(rgba, g["rgba"])


@pytest.mark.gpu
@pytest.mark.parametrize("name", sorted(CASES))
def test_hip_reproduces_golden(gpu_renderer, name):
    builder, w, h, seed = CASES[name]
    g = load(name)
    scene = builder()
    gpu_renderer.upload_scene(scene)
    rgba, mean = gpu_renderer.render(w, h, seed=seed, want_rgba8=True, want_mean=True)
    st = gpu_renderer.stats()
    c = compare(mean, rgba, g["mean"], g["rgba"])
    assert st.segments == int(g["segments"]), name
    assert max(c["rel_l2"]) <= 1e-4 and c["max_u8_delta"] <= 1, (name, c)
    assert c["mismatched_pixels"] == 0, (name, c)
