"""raytracingrust_amd -- MI355X-native (gfx950) path-tracing hot loop behind RayTracingRust's
`render(&mut RenderTarget, &Scene)` surface.  The numerical path is the hand-written HIP library
(csrc/, C ABI in include/rtr_hip.h); this package is the host-side mirror of the reference's scene API."""
from .scene import (Camera, Dielectric, Emission, GradientBackground, Isotropic, Lambertian, MaterialHandle, Mesh,
                    Metal, MixMaterial, ObjectHandle, RenderMode, RenderSettings, Scene, SkyMap, Sphere,
                    UniformBackground, Volume, load_obj)
from .render import DEFAULT_SEED, Renderer, RenderTarget, render
from ._capi import RtrError

__all__ = [
    "Camera", "Dielectric", "Emission", "GradientBackground", "Isotropic", "Lambertian", "MaterialHandle", "Mesh",
    "Metal", "MixMaterial", "ObjectHandle", "RenderMode", "RenderSettings", "Scene", "SkyMap", "Sphere",
    "UniformBackground", "Volume", "load_obj", "DEFAULT_SEED", "Renderer", "RenderTarget", "render", "RtrError",
]
