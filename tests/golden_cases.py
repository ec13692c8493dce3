"""The golden regression cases: name -> (scene builder, width, height, seed).  Shared by
tests/golden/make_golden.py (writer, oracle) and tests/test_golden.py (readers: oracle on CPU, HIP on GPU)."""
import numpy as np

import raytracingrust_amd as rt
from raytracingrust_amd.hdri import synthetic_sky
from helpers import benchmark_scene, house_scene, monkey_scene, simple_scene


def _sky_small():
    return rt.SkyMap(synthetic_sky(64, 32, sun_radius=0.08))


def _simple_sky():
    s = simple_scene(2)
    s.background = _sky_small()
    return s


def _mix_scene():
    """every material kind incl. nested MixMaterial and Isotropic, gradient background, linear world"""
    s = rt.Scene()
    s.settings.samples_per_pixel = 3
    s.settings.enable_bvh_tree = False
    s.background = rt.GradientBackground([1.0, 1.0, 1.0], [0.5, 0.7, 1.0])
    mats = [rt.Lambertian([0.7, 0.3, 0.3]), rt.Metal([0.8, 0.8, 0.8], 0.3), rt.Dielectric(1.5),
            rt.Emission([1.0, 0.9, 0.7], 4.0), rt.Isotropic([0.9, 0.9, 0.9]),
            rt.MixMaterial(rt.Metal([0.9, 0.6, 0.2], 0.0),
                           rt.MixMaterial(rt.Lambertian([0.2, 0.8, 0.2]), rt.Dielectric(1.3), 0.3), 0.6)]
    hs = [s.add_material(m) for m in mats]
    s.add_object(rt.Sphere([0.0, -100.5, -1.0], 100.0, hs[5]))
    for i, h in enumerate(hs[:5]):
        s.add_object(rt.Sphere([-1.6 + 0.8 * i, 0.0, -1.5], 0.38, h))
    return s


def _fog_scene():
    s = rt.Scene()
    s.settings.samples_per_pixel = 3
    s.settings.max_ray_depth = 8
    s.background = rt.GradientBackground([1.0, 1.0, 1.0], [0.5, 0.7, 1.0])
    ground = s.add_material(rt.Lambertian([0.5, 0.5, 0.5]))
    red = s.add_material(rt.Lambertian([0.8, 0.2, 0.2]))
    fog = s.add_material(rt.Isotropic([1.0, 1.0, 1.0]))
    smoke = s.add_material(rt.Isotropic([0.1, 0.1, 0.1]))
    s.add_object(rt.Sphere([0.0, -100.5, -1.0], 100.0, ground))
    s.add_object(rt.Sphere([0.6, 0.0, -1.2], 0.5, red))
    s.add_object(rt.Volume(rt.Sphere([0.0, 0.0, 0.0], 30.0, fog), 0.08))
    s.add_object(rt.Volume(rt.Sphere([-0.7, 0.2, -1.0], 0.45, smoke), 6.0))
    return s


def _mesh_fog_scene():
    """Volume over a mesh boundary (volume.rs:20): smoke inside cube.obj, next to a sphere and a sphere-bounded mist."""
    from raytracingrust_amd.configs import mesh_path

    s = rt.Scene()
    s.camera = rt.Camera([0.3, 0.6, 3.2], [0.0, 0.0, 0.0], [0.0, 1.0, 0.0], 55.0, 1.5)
    s.settings.samples_per_pixel = 2
    s.settings.max_ray_depth = 8
    s.background = rt.GradientBackground([1.0, 1.0, 1.0], [0.5, 0.7, 1.0])
    ground = s.add_material(rt.Lambertian([0.5, 0.5, 0.5]))
    red = s.add_material(rt.Lambertian([0.8, 0.2, 0.2]))
    smoke = s.add_material(rt.Isotropic([0.15, 0.15, 0.2]))
    mist = s.add_material(rt.Isotropic([0.9, 0.9, 0.9]))
    s.add_object(rt.Sphere([0.0, -101.0, 0.0], 100.0, ground))
    cube = rt.Mesh.from_file(mesh_path("cube.obj"), smoke)
    cube.positions = (cube.positions * np.float32(0.8) + np.array([-0.5, -0.1, 0.0], dtype=np.float32)).astype(np.float32)
    s.add_object(rt.Volume(cube, 3.0))
    s.add_object(rt.Sphere([1.1, -0.5, -0.3], 0.5, red))
    s.add_object(rt.Volume(rt.Sphere([0.0, 0.0, 0.0], 25.0, mist), 0.05))
    return s


CASES = {
    "benchmark_linear_64x36x4": (lambda: benchmark_scene(4), 64, 36, 0x5EED),
    "simple_bvh_48x48x2": (lambda: simple_scene(2), 48, 48, 17),
    "simple_sky_48x48x2": (_simple_sky, 48, 48, 18),
    "monkey_bvh_48x27x2": (lambda: monkey_scene(2), 48, 27, 19),
    "monkey_glass_sky_40x24x2": (lambda: monkey_scene(2, material=rt.Dielectric(1.5), background=_sky_small()), 40, 24, 20),
    "house_bvh_48x27x1": (lambda: house_scene(1), 48, 27, 21),
    "mix_linear_64x36x3": (_mix_scene, 64, 36, 22),
    "fog_volumes_bvh_48x32x3": (_fog_scene, 48, 32, 24),
    "mesh_fog_bvh_48x32x2": (_mesh_fog_scene, 48, 32, 25),
    "clay_mode_40x40x2": (lambda: simple_scene(2, mode=rt.RenderMode.Clay), 40, 40, 23),
}
