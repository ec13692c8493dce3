"""Scene builders for BASELINE.json's configurations (SURVEY.md section 8d gives the concrete definitions).
Pure scene construction through the public surface; used by bench.py, scripts/ and tests/."""
from __future__ import annotations

import os

import numpy as np

from .hdri import synthetic_sky
from .scene import (Camera, Dielectric, Emission, GradientBackground, Isotropic, Lambertian, Mesh, Metal, MixMaterial,
                    Scene, SkyMap)

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DATA = os.path.join(ROOT, "data")


def scene_path(name: str) -> str:
    return os.path.join(DATA, "scenes", name)


def mesh_path(name: str) -> str:
    return os.path.join(DATA, "meshes", name)


def benchmark_scene(spp=8, **settings) -> Scene:
    """C2: scenes/benchmark.json verbatim (5 spheres, linear world, aspect 1.2 kept)."""
    s = Scene.from_file(scene_path("benchmark.json"))
    s.settings.samples_per_pixel = spp
    for k, v in settings.items():
        setattr(s.settings, k, v)
    return s


def simple_scene(spp=8, **settings) -> Scene:
    """C1: scenes/simple.json verbatim (102 spheres, BVH world)."""
    s = Scene.from_file(scene_path("simple.json"))
    s.settings.samples_per_pixel = spp
    for k, v in settings.items():
        setattr(s.settings, k, v)
    return s


def monkey_scene(spp=4, material=None, background=None, depth=6, bvh=True, with_plane=True) -> Scene:
    """C3 geometry: monkey.obj over plane.obj, camera (0,0,3)->(0,0,0), fov 60, 16:9 (our definition: the
    reference's only monkey camera is the inside-the-mesh one of benches/mesh.rs:10-17)."""
    s = Scene()
    s.camera = Camera([0.0, 0.0, 3.0], [0.0, 0.0, 0.0], [0.0, 1.0, 0.0], 60.0, 16.0 / 9.0)
    s.settings.samples_per_pixel = spp
    s.settings.max_ray_depth = depth
    s.settings.enable_bvh_tree = bvh
    s.background = background if background is not None else GradientBackground([0.9, 0.9, 1.0], [0.3, 0.4, 0.8])
    m = s.add_material(material if material is not None else Lambertian([0.8, 0.5, 0.3]))
    g = s.add_material(Lambertian([0.4, 0.4, 0.4]))
    mesh = Mesh.from_file(mesh_path("monkey.obj"), m)
    mesh.build_bvh()
    s.add_object(mesh)
    if with_plane:
        plane = Mesh.from_file(mesh_path("plane.obj"), g)
        plane.positions = (plane.positions + np.array([0.0, -1.0, 0.0], dtype=np.float32)).astype(np.float32)
        s.add_object(plane)
    return s


def house_scene(spp=2, depth=3, background=None) -> Scene:
    """C4: the scene src/gui.rs:56-119 builds (house.obj + plane.obj, mix-material ground, depth 3)."""
    s = Scene()
    s.camera = Camera([-3.0, 2.0, 4.0], [0.0, 2.0, -1.0], [0.0, 1.0, 0.0], 90.0, 16.0 / 9.0)
    s.settings.samples_per_pixel = spp
    s.settings.max_ray_depth = depth
    s.background = background if background is not None else GradientBackground([1.0, 1.0, 1.0], [0.5, 0.7, 1.0])
    lam = s.add_material(Lambertian([0.9, 0.2, 0.2]))
    s.add_material(Metal([0.8, 0.2, 0.2], 0.02))
    s.add_material(Dielectric(1.5))
    ground = s.add_material(MixMaterial(Metal([0.1, 0.1, 0.1], 0.0), Lambertian([0.4, 0.4, 0.4]), 0.5))
    s.add_material(Emission([1.0, 1.0, 1.0], 50.0))
    s.add_material(Isotropic([1.0, 1.0, 1.0]))
    house = Mesh.from_file(mesh_path("house.obj"), lam)
    house.build_bvh()
    s.add_object(house)
    plane = Mesh.from_file(mesh_path("plane.obj"), ground)
    plane.build_bvh()
    s.add_object(plane)
    return s


def baseline_config(name: str, spp: int):
    """(scene, width, height) of a BASELINE.json configuration at `spp` samples per pixel."""
    def sky():
        return SkyMap(synthetic_sky(2048, 1024))

    if name == "c1":
        return simple_scene(spp), 1920, 1080
    if name == "c2":
        return benchmark_scene(spp), 1920, 1080
    if name == "c2b":  # benchmark.json with the BVH switched on
        return benchmark_scene(spp, enable_bvh_tree=True), 1920, 1080
    if name == "c3":   # glass monkey + HDRI
        return monkey_scene(spp, material=Dielectric(1.5), background=sky()), 1920, 1080
    if name == "c3l":  # Lambertian monkey + HDRI
        return monkey_scene(spp, background=sky()), 1920, 1080
    if name == "c4":
        return house_scene(spp, depth=3, background=sky()), 1920, 1080
    if name == "c5":
        s = simple_scene(spp)
        s.background = sky()
        return s, 3840, 2160
    raise ValueError(f"unknown config {name!r}")
