"""`render(target, scene)` -- the reference's entry point (lib/core/render.rs:61) over the HIP C ABI.

    RenderTarget      lib/core/render.rs:14-30  (width, height, RGBA8 `data`, request_redraw)
    render            lib/core/render.rs:61-127
    Renderer          the amortised form: one context + one uploaded scene, many frames

Everything numerical happens in librtr_hip.so; this module marshals arguments and owns buffers.
"""
from __future__ import annotations

import ctypes as C
from typing import Optional, Tuple

import numpy as np

from . import _capi
from .scene import LoweredScene, Scene, camera_struct, settings_struct

DEFAULT_SEED = 0x5EED5EED5EED5EED


class RenderTarget:
    """lib/core/render.rs:14-30"""

    def __init__(self, width: int, height: int):
        self.width = int(width)
        self.height = int(height)
        self.data = np.zeros(self.width * self.height * 4, dtype=np.uint8)
        self.request_redraw = False

    def image(self) -> np.ndarray:
        return self.data.reshape(self.height, self.width, 4)

    def save(self, path: str) -> None:
        """Writes the RGBA8 buffer as an image file (the reference only shows it in its egui
        window, src/gui.rs): ``.png`` -> 8-bit RGBA PNG, anything else -> binary PPM (P6, rgb)."""
        img = self.image()
        if not path.lower().endswith(".png"):
            with open(path, "wb") as f:
                f.write(b"P6\n%d %d\n255\n" % (self.width, self.height))
                f.write(np.ascontiguousarray(img[:, :, :3]).tobytes())
            return
        import struct
        import zlib

        def chunk(kind: bytes, body: bytes) -> bytes:
            return struct.pack(">I", len(body)) + kind + body + struct.pack(">I", zlib.crc32(kind + body) & 0xFFFFFFFF)

        raw = np.zeros((self.height, 1 + 4 * self.width), dtype=np.uint8)  # each scanline led by filter type 0
        raw[:, 1:] = img.reshape(self.height, 4 * self.width)
        with open(path, "wb") as f:
            f.write(b"\x89PNG\r\n\x1a\n")
            f.write(chunk(b"IHDR", struct.pack(">IIBBBBB", self.width, self.height, 8, 6, 0, 0, 0)))
            f.write(chunk(b"IDAT", zlib.compress(raw.tobytes(), 6)))
            f.write(chunk(b"IEND", b""))


class SceneInfo:
    def __init__(self, s: _capi.rtr_scene_info):
        for name, _ in s._fields_:
            setattr(self, name, int(getattr(s, name)))


class Stats:
    def __init__(self, s: _capi.rtr_stats):
        for name, _ in s._fields_:
            setattr(self, name, getattr(s, name))


class Renderer:
    """One rtr_context bound to one GPU (device >= 0) or host-only (device = -1: scene lowering + BVH
    introspection only; rendering raises RTR_ERR_NO_DEVICE -- there is no CPU fallback)."""

    def __init__(self, device=0):
        """`device`: a GPU index, -1 (host-only), or a sequence of GPU indices -- one context over several GPUs of this
        node (rtr_create_multi: frames are divided among them and reduced with RCCL inside the library)."""
        self._lib = _capi.load()
        self._ctx = C.c_void_p()
        if isinstance(device, (list, tuple)):
            ids = (C.c_int * len(device))(*[int(d) for d in device])
            rc = self._lib.rtr_create_multi(ids, len(device), C.byref(self._ctx))
        else:
            rc = self._lib.rtr_create(int(device), C.byref(self._ctx))
        if rc != _capi.RTR_OK:
            raise _capi.RtrError(rc, (self._lib.rtr_last_error(None) or b"").decode())
        self.device = device
        self._scene: Optional[Scene] = None

    def close(self) -> None:
        if getattr(self, "_ctx", None) is not None and self._ctx:
            self._lib.rtr_destroy(self._ctx)
            self._ctx = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def _check(self, rc: int) -> None:
        if rc != _capi.RTR_OK:
            raise _capi.RtrError(rc, (self._lib.rtr_last_error(self._ctx) or b"").decode())

    def set_option(self, name: str, value: str) -> None:
        """Context option, e.g. ("mesh_tree", "sah"); applied at the next upload_scene."""
        self._check(self._lib.rtr_set_option(self._ctx, name.encode(), value.encode()))

    # -- scene -------------------------------------------------------------------------------------------
    def upload_scene(self, scene: Scene) -> None:
        lowered = LoweredScene(scene)
        self._check(self._lib.rtr_upload_scene(self._ctx, C.byref(lowered.desc)))
        self._scene = scene

    def scene_info(self) -> SceneInfo:
        s = _capi.rtr_scene_info()
        self._check(self._lib.rtr_scene_info_get(self._ctx, C.byref(s)))
        return SceneInfo(s)

    def bvh_leaf_order(self) -> np.ndarray:
        """(object index, triangle handle) of every leaf in the reference's depth-first order."""
        n = self._lib.rtr_bvh_leaf_order(self._ctx, None, 0)
        if n < 0:
            self._check(int(n))
        out = np.zeros(max(int(n), 1), dtype=np.uint64)
        self._lib.rtr_bvh_leaf_order(self._ctx, out.ctypes.data_as(C.POINTER(C.c_uint64)), int(n))
        out = out[: int(n)]
        return np.stack([(out >> np.uint64(32)).astype(np.uint32), (out & np.uint64(0xFFFFFFFF)).astype(np.uint32)], axis=1)

    # -- frames ------------------------------------------------------------------------------------------
    def _cam_settings(self, scene: Optional[Scene]):
        sc = scene if scene is not None else self._scene
        if sc is None:
            raise _capi.RtrError(-7, "no scene uploaded")
        return camera_struct(sc.camera), settings_struct(sc.settings)

    def render(self, width: int, height: int, seed: int = DEFAULT_SEED, want_rgba8: bool = True,
               want_mean: bool = False, scene: Optional[Scene] = None) -> Tuple[Optional[np.ndarray], Optional[np.ndarray]]:
        """One frame with host outputs: (rgba8[h,w,4] | None, mean_rgb[h,w,3] | None).  `scene` overrides the
        camera/settings of the uploaded scene (its objects are NOT re-uploaded)."""
        cam, st = self._cam_settings(scene)
        rgba = np.zeros((height, width, 4), dtype=np.uint8) if want_rgba8 else None
        mean = np.zeros((height, width, 3), dtype=np.float32) if want_mean else None
        self._check(self._lib.rtr_render(
            self._ctx, C.byref(cam), C.byref(st), width, height, C.c_uint64(seed),
            rgba.ctypes.data_as(C.c_void_p) if rgba is not None else None,
            mean.ctypes.data_as(C.c_void_p) if mean is not None else None))
        return rgba, mean

    def render_frame_device(self, width: int, height: int, seed: int, d_rgba8_ptr: int, stream: int = 0,
                            scene: Optional[Scene] = None) -> None:
        cam, st = self._cam_settings(scene)
        self._check(self._lib.rtr_render_frame_device(self._ctx, C.byref(cam), C.byref(st), width, height,
                                                      C.c_uint64(seed), C.c_void_p(d_rgba8_ptr), C.c_void_p(stream)))

    def render_tiles_device(self, width: int, height: int, seed: int, shard: int, n_shards: int, d_sum_ptr: int,
                            stream: int = 0, scene: Optional[Scene] = None) -> None:
        cam, st = self._cam_settings(scene)
        self._check(self._lib.rtr_render_tiles_device(self._ctx, C.byref(cam), C.byref(st), width, height,
                                                      C.c_uint64(seed), shard, n_shards, C.c_void_p(d_sum_ptr),
                                                      C.c_void_p(stream)))

    def render_samples_device(self, width: int, height: int, seed: int, sample_begin: int, sample_count: int,
                              d_sum_ptr: int, stream: int = 0, scene: Optional[Scene] = None) -> None:
        cam, st = self._cam_settings(scene)
        self._check(self._lib.rtr_render_samples_device(self._ctx, C.byref(cam), C.byref(st), width, height,
                                                        C.c_uint64(seed), sample_begin, sample_count,
                                                        C.c_void_p(d_sum_ptr), C.c_void_p(stream)))

    def resolve_device(self, d_sum_ptr: int, spp: int, width: int, height: int, d_rgba8_ptr: int = 0,
                       d_mean_ptr: int = 0, stream: int = 0) -> None:
        self._check(self._lib.rtr_resolve_device(self._ctx, C.c_void_p(d_sum_ptr), spp, width, height,
                                                 C.c_void_p(d_rgba8_ptr), C.c_void_p(d_mean_ptr), C.c_void_p(stream)))

    def stats(self) -> Stats:
        s = _capi.rtr_stats()
        self._check(self._lib.rtr_get_stats(self._ctx, C.byref(s)))
        return Stats(s)

    def kernel_times_ms(self, cap: int = 64) -> np.ndarray:
        out = np.zeros(cap, dtype=np.float32)
        n = self._lib.rtr_kernel_times_ms(self._ctx, out.ctypes.data_as(C.POINTER(C.c_float)), cap)
        if n < 0:
            self._check(n)
        return out[:n]

    def integrate_times_ms(self, cap: int = 64) -> np.ndarray:
        """Per launch: start of the launch to the end of its last integrate kernel (the dominant kernel alone)."""
        out = np.zeros(cap, dtype=np.float32)
        n = self._lib.rtr_integrate_times_ms(self._ctx, out.ctypes.data_as(C.POINTER(C.c_float)), cap)
        if n < 0:
            self._check(n)
        return out[:n]

    def kernel_name(self) -> str:
        return (self._lib.rtr_kernel_name(self._ctx) or b"").decode()

    def selftest_sqrt(self) -> int:
        """Mismatches between the kernels' sqrt and sqrtf() over all 2^32 f32 bit patterns (must be 0)."""
        bad = C.c_uint64(0)
        self._check(self._lib.rtr_selftest_sqrt(self._ctx, C.byref(bad)))
        return int(bad.value)

    def selftest_div(self, denominator: float, lo: float, hi: float) -> int:
        """Mismatches between the kernels' shared-denominator division and n / denominator over every f32 n in
        [lo, hi] (0 <= lo <= hi, or lo <= hi <= -0.0 handled by the caller's choice of bit range); must be 0."""
        lo_bits = int(np.float32(lo).view(np.uint32))
        hi_bits = int(np.float32(hi).view(np.uint32))
        bad = C.c_uint64(0)
        self._check(self._lib.rtr_selftest_div(self._ctx, C.c_float(denominator), lo_bits, hi_bits, C.byref(bad)))
        return int(bad.value)

    def probe_hits(self, origins, directions, enable_bvh_tree: Optional[bool] = None, seed: int = 0):
        """world.hit(ray, 0.00001, inf) (render.rs:38) of the uploaded scene for caller-supplied rays, on the device.
        Returns one dict per ray (t, point, normal, front_face, material) or None for a miss."""
        o = np.ascontiguousarray(origins, dtype=np.float32).reshape(-1, 3)
        d = np.ascontiguousarray(directions, dtype=np.float32).reshape(-1, 3)
        assert o.shape == d.shape
        if enable_bvh_tree is None:
            enable_bvh_tree = bool(self._scene.settings.enable_bvh_tree) if self._scene is not None else True
        out = (_capi.rtr_hit * max(len(o), 1))()
        self._check(self._lib.rtr_probe_hits(self._ctx, 1 if enable_bvh_tree else 0, C.c_uint64(seed), len(o),
                                             o.ctypes.data_as(C.c_void_p), d.ctypes.data_as(C.c_void_p), out))
        res = []
        for i in range(len(o)):
            h = out[i]
            res.append(None if not h.hit else {
                "t": float(np.float32(h.t)), "point": np.array(list(h.point), dtype=np.float32),
                "normal": np.array(list(h.normal), dtype=np.float32), "front_face": bool(h.front_face),
                "material": int(h.material)})
        return res

    def segment_counts(self, cap: int = 64) -> np.ndarray:
        out = np.zeros(cap, dtype=np.uint64)
        n = self._lib.rtr_segment_counts(self._ctx, out.ctypes.data_as(C.POINTER(C.c_uint64)), cap)
        if n < 0:
            self._check(n)
        return out[:n]


def camera_ray_origin(camera) -> np.ndarray:
    """Camera::ray_origin (lib/core/camera.rs:37-56) as the library computes it: rows origin, horizontal, vertical,
    lower_left_corner.  Host arithmetic; needs no device."""
    out = (C.c_float * 12)()
    rc = _capi.load().rtr_camera_ray_origin(C.byref(camera_struct(camera)), out)
    if rc != _capi.RTR_OK:
        raise _capi.RtrError(rc, "rtr_camera_ray_origin")
    return np.array(list(out), dtype=np.float32).reshape(4, 3)


def render(target: RenderTarget, scene: Scene, seed: int = DEFAULT_SEED, device: int = 0) -> None:
    """lib/core/render.rs:61: fills `target.data` (RGBA8, row-major, y = 0 on top) and sets request_redraw.
    Like the reference, the scene's acceleration structure is rebuilt on every call (render.rs:67); use a
    `Renderer` to amortise the upload over many frames."""
    target.request_redraw = True
    with Renderer(device) as r:
        r.upload_scene(scene)
        rgba, _ = r.render(target.width, target.height, seed=seed, want_rgba8=True)
    target.data[:] = rgba.reshape(-1)
