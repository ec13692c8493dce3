"""ctypes binding of include/rtr_hip.h.  Loading fails loudly when the HIP library is missing:
there is no Python or CPU fallback for the render path."""
from __future__ import annotations

import ctypes as C
import os

from . import build as _build

c_float3 = C.c_float * 3


class rtr_sphere(C.Structure):
    _fields_ = [("center", c_float3), ("radius", C.c_float), ("material", C.c_uint32)]


class rtr_mesh(C.Structure):
    _fields_ = [("positions", C.POINTER(C.c_float)), ("normals", C.POINTER(C.c_float)),
                ("indices", C.POINTER(C.c_uint32)), ("n_verts", C.c_uint32), ("n_tris", C.c_uint32),
                ("material", C.c_uint32), ("reserved", C.c_uint32)]


class rtr_volume(C.Structure):
    _fields_ = [("boundary", rtr_sphere), ("neg_inv_density", C.c_float), ("boundary_kind", C.c_uint32),
                ("boundary_mesh", C.c_uint32)]


class rtr_material(C.Structure):
    _fields_ = [("kind", C.c_uint32), ("rgb", c_float3), ("param", C.c_float), ("first", C.c_uint32),
                ("second", C.c_uint32)]


class rtr_object(C.Structure):
    _fields_ = [("kind", C.c_uint32), ("index", C.c_uint32)]


class rtr_background(C.Structure):
    _fields_ = [("kind", C.c_uint32), ("a", c_float3), ("b", c_float3), ("image", C.POINTER(C.c_float)),
                ("width", C.c_uint32), ("height", C.c_uint32)]


class rtr_scene_desc(C.Structure):
    _fields_ = [("objects", C.POINTER(rtr_object)), ("n_objects", C.c_uint32),
                ("spheres", C.POINTER(rtr_sphere)), ("n_spheres", C.c_uint32),
                ("meshes", C.POINTER(rtr_mesh)), ("n_meshes", C.c_uint32),
                ("materials", C.POINTER(rtr_material)), ("n_materials", C.c_uint32),
                ("background", rtr_background),
                ("volumes", C.POINTER(rtr_volume)), ("n_volumes", C.c_uint32)]


class rtr_camera(C.Structure):
    _fields_ = [("lookfrom", c_float3), ("lookat", c_float3), ("vertical", c_float3),
                ("vertical_fov", C.c_float), ("aspect_ratio", C.c_float)]


class rtr_settings(C.Structure):
    _fields_ = [("samples_per_pixel", C.c_uint32), ("max_ray_depth", C.c_uint32), ("clamp_indirect", C.c_float),
                ("enable_multithreading", C.c_uint32), ("enable_bvh_tree", C.c_uint32), ("mode", C.c_uint32)]


class rtr_scene_info(C.Structure):
    _fields_ = [("n_objects", C.c_uint32), ("n_spheres", C.c_uint32), ("n_meshes", C.c_uint32),
                ("n_triangles", C.c_uint32), ("n_materials", C.c_uint32), ("n_leaves", C.c_uint32),
                ("n_nodes", C.c_uint32), ("tree_height", C.c_uint32), ("spheres_only", C.c_uint32),
                ("device_bytes", C.c_uint64)]


class rtr_stats(C.Structure):
    _fields_ = [("primary_samples", C.c_uint64), ("segments", C.c_uint64), ("kernel_ms", C.c_double),
                ("resolve_ms", C.c_double), ("upload_ms", C.c_double), ("launches", C.c_uint32),
                ("vgprs", C.c_uint32), ("lds_bytes", C.c_uint32), ("block_threads", C.c_uint32),
                ("node_steps", C.c_uint64), ("leaf_tests", C.c_uint64), ("sky_lookups", C.c_uint64),
                ("traversal_counted", C.c_uint32), ("reserved", C.c_uint32)]


class rtr_hit(C.Structure):
    _fields_ = [("hit", C.c_uint32), ("t", C.c_float), ("point", c_float3), ("normal", c_float3),
                ("front_face", C.c_uint32), ("material", C.c_uint32), ("reserved", C.c_uint32 * 2)]


RTR_OK = 0
STATUS_NAMES = {
    0: "RTR_OK", -1: "RTR_ERR_INVALID_ARGUMENT", -2: "RTR_ERR_NO_DEVICE", -3: "RTR_ERR_HIP", -4: "RTR_ERR_BAD_INDEX",
    -5: "RTR_ERR_NAN_CENTROID", -6: "RTR_ERR_BAD_SETTINGS", -7: "RTR_ERR_NO_SCENE", -8: "RTR_ERR_UNSUPPORTED",
    -9: "RTR_ERR_OUT_OF_MEMORY",
}

# every symbol include/rtr_hip.h declares: (name, restype, argtypes)
_ctx = C.c_void_p
SYMBOLS = [
    ("rtr_abi_version", C.c_int, []),
    ("rtr_create", C.c_int, [C.c_int, C.POINTER(_ctx)]),
    ("rtr_create_multi", C.c_int, [C.POINTER(C.c_int), C.c_int, C.POINTER(_ctx)]),
    ("rtr_sample_range", C.c_int, [C.c_uint32, C.c_uint32, C.c_uint32, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]),
    ("rtr_destroy", None, [_ctx]),
    ("rtr_last_error", C.c_char_p, [_ctx]),
    ("rtr_set_option", C.c_int, [_ctx, C.c_char_p, C.c_char_p]),
    ("rtr_upload_scene", C.c_int, [_ctx, C.POINTER(rtr_scene_desc)]),
    ("rtr_scene_info_get", C.c_int, [_ctx, C.POINTER(rtr_scene_info)]),
    ("rtr_bvh_leaf_order", C.c_int64, [_ctx, C.POINTER(C.c_uint64), C.c_uint64]),
    ("rtr_render", C.c_int, [_ctx, C.POINTER(rtr_camera), C.POINTER(rtr_settings), C.c_uint32, C.c_uint32,
                             C.c_uint64, C.c_void_p, C.c_void_p]),
    ("rtr_render_rgba8", C.c_int, [_ctx, C.POINTER(rtr_camera), C.POINTER(rtr_settings), C.c_uint32, C.c_uint32,
                                   C.c_uint64, C.c_void_p]),
    ("rtr_render_f32", C.c_int, [_ctx, C.POINTER(rtr_camera), C.POINTER(rtr_settings), C.c_uint32, C.c_uint32,
                                 C.c_uint64, C.c_void_p]),
    ("rtr_render_frame_device", C.c_int, [_ctx, C.POINTER(rtr_camera), C.POINTER(rtr_settings), C.c_uint32,
                                          C.c_uint32, C.c_uint64, C.c_void_p, C.c_void_p]),
    ("rtr_render_tiles_device", C.c_int, [_ctx, C.POINTER(rtr_camera), C.POINTER(rtr_settings), C.c_uint32,
                                          C.c_uint32, C.c_uint64, C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p]),
    ("rtr_render_samples_device", C.c_int, [_ctx, C.POINTER(rtr_camera), C.POINTER(rtr_settings), C.c_uint32,
                                            C.c_uint32, C.c_uint64, C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p]),
    ("rtr_resolve_device", C.c_int, [_ctx, C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p,
                                     C.c_void_p]),
    ("rtr_get_stats", C.c_int, [_ctx, C.POINTER(rtr_stats)]),
    ("rtr_kernel_times_ms", C.c_int, [_ctx, C.POINTER(C.c_float), C.c_uint32]),
    ("rtr_segment_counts", C.c_int, [_ctx, C.POINTER(C.c_uint64), C.c_uint32]),
    ("rtr_integrate_times_ms", C.c_int, [_ctx, C.POINTER(C.c_float), C.c_uint32]),
    ("rtr_kernel_name", C.c_char_p, [_ctx]),
    ("rtr_selftest_sqrt", C.c_int, [_ctx, C.POINTER(C.c_uint64)]),
    ("rtr_selftest_div", C.c_int, [_ctx, C.c_float, C.c_uint32, C.c_uint32, C.POINTER(C.c_uint64)]),
    ("rtr_probe_hits", C.c_int, [_ctx, C.c_uint32, C.c_uint64, C.c_uint32, C.c_void_p, C.c_void_p, C.POINTER(rtr_hit)]),
    ("rtr_camera_ray_origin", C.c_int, [C.POINTER(rtr_camera), C.POINTER(C.c_float)]),
]

_lib = None


class RtrError(RuntimeError):
    def __init__(self, code: int, message: str):
        super().__init__(f"{STATUS_NAMES.get(code, code)}: {message}")
        self.code = code


def library_path() -> str:
    # RTR_HIP_LIBRARY: point at another build of the SAME library (kernel experiments); never a fallback
    return os.environ.get("RTR_HIP_LIBRARY") or _build.LIB_PATH


def load():
    """Load librtr_hip.so (no auto-build: run `python -m raytracingrust_amd.build` / __graft_entry__.build())."""
    global _lib
    if _lib is not None:
        return _lib
    path = library_path()
    if not os.path.exists(path):
        raise ImportError(
            f"{path} is missing: the HIP extension has not been built (python -m raytracingrust_amd.build). "
            "raytracingrust_amd has no CPU fallback.")
    # PyTorch-ROCm wheels bundle their own HIP/HSA runtime.  Whichever of torch and this library is loaded first
    # decides which copy the process initialises, and torch finds no GPU when ours came first -- so when torch is
    # installed (it carries the streams / RCCL plumbing of distributed.py and bench.py) it goes first.
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    lib = C.CDLL(path)
    for name, restype, argtypes in SYMBOLS:
        fn = getattr(lib, name)  # AttributeError if the library does not export a declared symbol
        fn.restype = restype
        fn.argtypes = argtypes
    _lib = lib
    return lib
