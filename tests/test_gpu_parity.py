"""GPU parity: the HIP path (through the C ABI) against the CPU oracle on identical seeds.

Tolerance (BASELINE.json north_star): float framebuffer within 1e-4 relative L2 per channel, RGBA8 within
+-1 LSB.  Because kernels and oracle execute the same f32 operation sequence (no contraction, correctly
rounded divide/sqrt), sphere scenes are additionally required to be BIT-IDENTICAL, and the device-counted
ray segments must equal the oracle's.
"""
import numpy as np
import pytest

import pyoracle
import raytracingrust_amd as rt
from helpers import benchmark_scene, compare, house_scene, monkey_scene, simple_scene

pytestmark = pytest.mark.gpu

TOL = 1e-4  # relative L2 per channel (BASELINE.json north_star)


def run_both(r, scene, w, h, seed, trig=pyoracle.TRIG_POLY):
    r.upload_scene(scene)
    rgba, mean = r.render(w, h, seed=seed, want_rgba8=True, want_mean=True)
    st = r.stats()
    o_rgba, o_mean, o_st = pyoracle.OracleScene(scene, trig_mode=trig).render(w, h, seed, threads=0)
    return compare(mean, rgba, o_mean, o_rgba), st, o_st


@pytest.mark.parametrize("w,h,spp,seed", [(160, 90, 16, 1), (97, 61, 5, 0xDEADBEEFCAFE), (8, 8, 64, 7)])
def test_benchmark_scene_linear_world_bit_exact(gpu_renderer, w, h, spp, seed):
    """BASELINE config C2's scene (scenes/benchmark.json, linear world, 5 spheres)."""
    c, st, o_st = run_both(gpu_renderer, benchmark_scene(spp), w, h, seed)
    assert st.segments == o_st.segments
    assert c["mismatched_pixels"] == 0, c
    assert c["max_u8_delta"] == 0


def test_benchmark_scene_bvh_world(gpu_renderer):
    c, st, o_st = run_both(gpu_renderer, benchmark_scene(8, enable_bvh_tree=True), 128, 72, 3)
    assert st.segments == o_st.segments
    assert c["mismatched_pixels"] == 0, c


@pytest.mark.parametrize("bvh", [True, False])
def test_simple_scene(gpu_renderer, bvh):
    """BASELINE config C1's scene (scenes/simple.json: 102 spheres, all four material kinds)."""
    c, st, o_st = run_both(gpu_renderer, simple_scene(8, enable_bvh_tree=bvh), 128, 128, 11)
    assert max(c["rel_l2"]) <= TOL, c
    assert c["max_u8_delta"] <= 1
    assert st.segments == o_st.segments
    assert c["mismatched_pixels"] == 0, c


@pytest.mark.parametrize("mode", [rt.RenderMode.Clay, rt.RenderMode.Normal, rt.RenderMode.Random])
def test_render_modes(gpu_renderer, mode):
    c, st, o_st = run_both(gpu_renderer, simple_scene(4, mode=mode), 96, 96, 5)
    assert st.segments == o_st.segments
    assert c["mismatched_pixels"] == 0, c


@pytest.mark.parametrize("depth", [0, 1, 2, 50])
def test_depth_limits(gpu_renderer, depth):
    c, st, o_st = run_both(gpu_renderer, benchmark_scene(4, max_ray_depth=depth), 64, 48, 9)
    assert st.segments == o_st.segments
    assert c["mismatched_pixels"] == 0, c


def test_monkey_mesh_bvh(gpu_renderer):
    """BASELINE config C3 geometry (monkey.obj, 967 triangles + plane), gradient background."""
    c, st, o_st = run_both(gpu_renderer, monkey_scene(4), 160, 90, 21)
    assert max(c["rel_l2"]) <= TOL, c
    assert c["max_u8_delta"] <= 1
    assert st.segments == o_st.segments
    assert c["mismatched_pixels"] <= 2, c  # culled traversal vs unculled: ulp-level ties only


def test_monkey_glass_inside_camera(gpu_renderer):
    """benches/mesh.rs: glass monkey, default camera at the origin (inside the mesh), depth 50."""
    s = monkey_scene(4, material=rt.Dielectric(1.5), depth=50, with_plane=False)
    s.camera = rt.Camera()
    c, st, o_st = run_both(gpu_renderer, s, 40, 40, 2)
    assert max(c["rel_l2"]) <= TOL, c
    assert st.segments == o_st.segments


def test_monkey_linear_world_with_meshes(gpu_renderer):
    c, st, o_st = run_both(gpu_renderer, monkey_scene(3, bvh=False), 96, 54, 4)
    assert max(c["rel_l2"]) <= TOL, c
    assert st.segments == o_st.segments


def test_house_scene(gpu_renderer):
    """BASELINE config C4 geometry (src/gui.rs:56-119: house.obj 9153 triangles + plane, mix material)."""
    c, st, o_st = run_both(gpu_renderer, house_scene(2), 160, 90, 6)
    assert max(c["rel_l2"]) <= TOL, c
    assert c["max_u8_delta"] <= 1
    assert st.segments == o_st.segments
