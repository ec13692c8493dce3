"""ctypes binding of the CPU oracle (oracle/rt_oracle.c).  TEST INFRASTRUCTURE ONLY: imported by tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg -- never by raytracingrust_amd/ (the product)."""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from typing import Optional, Tuple

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "librt_oracle.so")

TRIG_LIBM, TRIG_POLY = 0, 1


class or_stats(C.Structure):
    _fields_ = [("primary_samples", C.c_uint64), ("segments", C.c_uint64), ("box_tests", C.c_uint64),
                ("sphere_tests", C.c_uint64), ("tri_tests", C.c_uint64), ("sky_lookups", C.c_uint64),
                ("rng_draws", C.c_uint64), ("seconds", C.c_double), ("threads", C.c_int)]


def build(force: bool = False) -> str:
    src = [os.path.join(HERE, f) for f in ("rt_oracle.c", "rt_oracle.h", "Makefile")]
    if force or not os.path.exists(LIB_PATH) or any(os.path.getmtime(s) > os.path.getmtime(LIB_PATH) for s in src):
        subprocess.run(["make", "-C", HERE, "-B" if force else "-s", "librt_oracle.so"], check=True,
                       capture_output=True)
    return LIB_PATH


_lib = None
_f3 = C.c_float * 3


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            build()
        L = C.CDLL(LIB_PATH)
        L.or_scene_new.restype = C.c_void_p
        L.or_scene_free.argtypes = [C.c_void_p]
        L.or_set_camera.argtypes = [C.c_void_p, _f3, _f3, _f3, C.c_float, C.c_float]
        L.or_set_settings.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_float, C.c_int, C.c_int, C.c_int]
        L.or_set_trig_mode.argtypes = [C.c_void_p, C.c_int]
        L.or_set_sky_filter.argtypes = [C.c_void_p, C.c_int]
        L.or_set_background_uniform.argtypes = [C.c_void_p, _f3]
        L.or_set_background_gradient.argtypes = [C.c_void_p, _f3, _f3]
        L.or_set_background_skymap.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32]
        L.or_add_material.argtypes = [C.c_void_p, C.c_int, _f3, C.c_float, C.c_int, C.c_int]
        L.or_add_sphere.argtypes = [C.c_void_p, _f3, C.c_float, C.c_int]
        L.or_add_mesh.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32, C.c_int]
        L.or_add_volume_sphere.argtypes = [C.c_void_p, _f3, C.c_float, C.c_int, C.c_float]
        L.or_add_volume_mesh.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32, C.c_int, C.c_float]
        L.or_wrap_last_volume.argtypes = [C.c_void_p, C.c_float]
        L.or_render.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint64, C.c_int, C.c_uint32, C.c_uint32,
                                C.c_void_p, C.c_void_p, C.POINTER(or_stats)]
        L.or_camera_ray_origin.argtypes = [C.c_void_p, C.POINTER(C.c_float)]
        L.or_probe_hit.argtypes = [C.c_void_p, _f3, _f3, C.POINTER(C.c_float)]
        L.or_probe_aabb.argtypes = [_f3, _f3, _f3, _f3, C.c_float, C.c_float]
        L.or_probe_bvh_leaf_order.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
        L.or_probe_mesh_leaf_order.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int]
        L.or_probe_rng.argtypes = [C.c_uint64, C.c_uint32, C.c_uint32, C.c_int, C.c_void_p, C.c_int]
        L.or_probe_acos.argtypes = [C.c_float, C.c_int]
        L.or_probe_acos.restype = C.c_float
        L.or_probe_atan2.argtypes = [C.c_float, C.c_float, C.c_int]
        L.or_probe_atan2.restype = C.c_float
        L.or_probe_log.argtypes = [C.c_float, C.c_int]
        L.or_probe_log.restype = C.c_float
        L.or_last_error.restype = C.c_char_p
        _lib = L
    return _lib


def f3(v):
    return _f3(*[float(x) for x in v])


class OracleError(RuntimeError):
    pass


class OracleScene:
    """Builds an or_scene from a raytracingrust_amd.Scene (the same object the HIP path is fed)."""

    def __init__(self, scene, trig_mode: int = TRIG_LIBM, sky_bilinear: bool = False):
        from raytracingrust_amd import scene as S  # the scene *model* only: plain containers, no numerics
        L = lib()
        self._L = L
        self._s = C.c_void_p(L.or_scene_new())
        cam, st = scene.camera, scene.settings
        L.or_set_camera(self._s, f3(cam.lookfrom), f3(cam.lookat), f3(cam.vertical), float(np.float32(cam.vertical_fov)),
                        float(np.float32(cam.aspect_ratio)))
        L.or_set_settings(self._s, st.samples_per_pixel, st.max_ray_depth, float(np.float32(st.clamp_indirect)),
                          int(st.enable_multithreading), int(st.enable_bvh_tree), int(st.mode))
        L.or_set_trig_mode(self._s, trig_mode)
        L.or_set_sky_filter(self._s, 1 if sky_bilinear else 0)
        bg = scene.background
        if isinstance(bg, S.UniformBackground):
            L.or_set_background_uniform(self._s, f3(bg.color))
        elif isinstance(bg, S.GradientBackground):
            L.or_set_background_gradient(self._s, f3(bg.top), f3(bg.bottom))
        else:
            self._check(L.or_set_background_skymap(self._s, bg.image.ctypes.data_as(C.c_void_p), bg.width, bg.height))
        # materials: scene handles first, nested mix children appended (same flattening as the ABI)
        lowered = scene.lowered_materials()
        # children have larger indices than parents; the oracle wants children added first -> add in
        # reverse order into a remapped index space
        n = len(lowered)
        remap = {}
        for old in range(n - 1, -1, -1):
            m = lowered[old]
            first = remap[m.first] if m.kind == S.MAT_MIX else 0
            second = remap[m.second] if m.kind == S.MAT_MIX else 0
            new = L.or_add_material(self._s, m.kind, f3(m.rgb), float(np.float32(m.param)), first, second)
            self._check(new)
            remap[old] = new
        self.material_remap = remap
        for o in scene.objects:
            if isinstance(o, S.Sphere):
                self._check(L.or_add_sphere(self._s, f3(o.center), float(np.float32(o.radius)), remap.get(o.material, o.material)))
            elif isinstance(o, S.Mesh):
                self._check(L.or_add_mesh(self._s, o.positions.ctypes.data_as(C.c_void_p),
                                          o.normals.ctypes.data_as(C.c_void_p), o.positions.shape[0],
                                          o.indices.ctypes.data_as(C.c_void_p), o.n_tris,
                                          remap.get(o.material, o.material)))
            elif isinstance(o, S.Volume):
                chain = [o]  # outermost first; Volume::new(Box::new(inner), density) from the inside out
                while isinstance(chain[-1].boundary, S.Volume):
                    chain.append(chain[-1].boundary)
                o = chain[-1]
                b = o.boundary
                if isinstance(b, S.Mesh):
                    self._check(L.or_add_volume_mesh(self._s, b.positions.ctypes.data_as(C.c_void_p),
                                                     b.normals.ctypes.data_as(C.c_void_p), b.positions.shape[0],
                                                     b.indices.ctypes.data_as(C.c_void_p), b.n_tris,
                                                     remap.get(b.material, b.material), float(np.float32(o.density))))
                else:
                    self._check(L.or_add_volume_sphere(self._s, f3(b.center), float(np.float32(b.radius)),
                                                       remap.get(b.material, b.material), float(np.float32(o.density))))
                for outer in reversed(chain[:-1]):
                    self._check(L.or_wrap_last_volume(self._s, float(np.float32(outer.density))))
            else:
                raise TypeError(o)

    def _check(self, rc: int):
        if rc < 0:
            raise OracleError(f"oracle error {rc}: {(self._L.or_last_error() or b'').decode()}")
        return rc

    def add_volume_sphere(self, center, radius, material, density):
        return self._check(self._L.or_add_volume_sphere(self._s, f3(center), float(radius),
                                                        self.material_remap.get(material, material), float(density)))

    def set_settings(self, st):
        self._L.or_set_settings(self._s, st.samples_per_pixel, st.max_ray_depth, float(np.float32(st.clamp_indirect)),
                                int(st.enable_multithreading), int(st.enable_bvh_tree), int(st.mode))

    def render(self, width: int, height: int, seed: int, threads: int = 0, rows: Tuple[int, int] = (0, 0),
               want_rgba8: bool = True, want_mean: bool = True):
        rgba = np.zeros((height, width, 4), dtype=np.uint8) if want_rgba8 else None
        mean = np.zeros((height, width, 3), dtype=np.float32) if want_mean else None
        st = or_stats()
        self._check(self._L.or_render(self._s, width, height, C.c_uint64(seed), threads, rows[0], rows[1],
                                      rgba.ctypes.data_as(C.c_void_p) if rgba is not None else None,
                                      mean.ctypes.data_as(C.c_void_p) if mean is not None else None, C.byref(st)))
        return rgba, mean, st

    def camera_ray_origin(self) -> np.ndarray:
        out = (C.c_float * 12)()
        self._L.or_camera_ray_origin(self._s, out)
        return np.array(out, dtype=np.float32).reshape(4, 3)

    def probe_hit(self, origin, direction) -> Optional[dict]:
        out = (C.c_float * 9)()
        r = self._check(self._L.or_probe_hit(self._s, f3(origin), f3(direction), out))
        if not r:
            return None
        a = np.array(out, dtype=np.float32)
        return {"t": a[0], "point": a[1:4], "normal": a[4:7], "front_face": bool(a[7]),
                "material": int(a[8])}

    def bvh_leaf_order(self) -> np.ndarray:
        n = self._check(self._L.or_probe_bvh_leaf_order(self._s, None, 0))
        out = np.zeros(max(n, 1), dtype=np.uint32)
        self._L.or_probe_bvh_leaf_order(self._s, out.ctypes.data_as(C.c_void_p), n)
        return out[:n]

    def mesh_leaf_order(self, obj: int) -> np.ndarray:
        n = self._check(self._L.or_probe_mesh_leaf_order(self._s, obj, None, 0))
        out = np.zeros(max(n, 1), dtype=np.uint32)
        self._L.or_probe_mesh_leaf_order(self._s, obj, out.ctypes.data_as(C.c_void_p), n)
        return out[:n]

    def close(self):
        if self._s:
            self._L.or_scene_free(self._s)
            self._s = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def probe_aabb(bmin, bmax, origin, direction, tmin, tmax) -> bool:
    return bool(lib().or_probe_aabb(f3(bmin), f3(bmax), f3(origin), f3(direction), float(tmin), float(tmax)))


def probe_rng(seed: int, pixel: int, sample: int, kind: int, n: int) -> np.ndarray:
    out = np.zeros(n, dtype=np.float32)
    lib().or_probe_rng(C.c_uint64(seed), pixel, sample, kind, out.ctypes.data_as(C.c_void_p), n)
    return out
