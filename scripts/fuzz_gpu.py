"""Wider fuzz than tests/test_gpu_parity.py::test_random_scenes: N seeded random scenes (spheres, triangle soups, volumes, every
material and background, both worlds) under three tree kernels, each frame compared with the oracle bit for bit.
usage (GPU box): python scripts/fuzz_gpu.py N      -- 300 scenes take about 10 s."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np
import raytracingrust_amd as rt

import test_gpu_parity as T
r = rt.Renderer(0)
bad = 0
for case in range(200, 200 + int(sys.argv[1])):
    rng = np.random.default_rng(5000 + case)
    with_mesh, with_volume = case % 3 == 1, case % 3 == 2
    scene = T._random_scene(rng, with_mesh, with_volume)
    for tk in ("auto", "samples", "wave"):
        r.set_option("tree_kernel", tk)
        w, h = int(rng.integers(5, 70)), int(rng.integers(5, 50))
        c, st, o_st = T.run_both(r, scene, w, h, int(rng.integers(0, 2**62)))
        ok = st.segments == o_st.segments and c["mismatched_pixels"] == 0
        if not ok:
            bad += 1
            print("MISMATCH case", case, tk, st.segments, o_st.segments, c)
print("fuzz done, mismatches:", bad)
