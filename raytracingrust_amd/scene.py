"""Host-side mirror of the reference's scene surface (names, fields and defaults follow the Rust):

    Scene / RenderSettings / RenderMode      lib/core/scene.rs:14-115
    Camera                                   lib/core/camera.rs:27-34
    Sphere                                   lib/objects.rs:15-20
    Mesh (from_file / from_buffers)          lib/core/mesh.rs:62-117
    Lambertian, Metal, Emission, Dielectric,
    MixMaterial                              lib/materials.rs
    Volume, Isotropic                        lib/volume.rs:19-32,75-88
    UniformBackground, GradientBackground,
    SkyMap                                   lib/backgrounds.rs

Only plumbing lives here (containers, the scene-JSON reader of scenes/*.json, the OBJ reader, lowering
to the C ABI's plain-old-data).  Nothing in this file intersects, shades or builds a BVH: that is the
HIP library's job (csrc/).  All numeric fields are held as numpy float32, the reference's `Float`
(lib/utils/types.rs:4).
"""
from __future__ import annotations

import ctypes as C
import enum
import json
from dataclasses import dataclass, field
from typing import List, Optional, Sequence, Tuple, Union

import numpy as np

from . import _capi

f32 = np.float32


def _v3(v) -> np.ndarray:
    a = np.asarray(v, dtype=np.float32).reshape(3)
    return a.copy()


def _xyz(d) -> np.ndarray:
    return _v3([d["x"], d["y"], d["z"]])


def _rgb(d) -> np.ndarray:
    return _v3([d["r"], d["g"], d["b"]])


class RenderMode(enum.IntEnum):
    """lib/core/scene.rs:14-20"""
    Full = 0
    Clay = 1
    Random = 2
    Normal = 3


@dataclass
class RenderSettings:
    """lib/core/scene.rs:22-30; defaults scene.rs:52-59"""
    samples_per_pixel: int = 5
    max_ray_depth: int = 6
    clamp_indirect: float = 10.0
    enable_multithreading: bool = True
    enable_bvh_tree: bool = True
    mode: RenderMode = RenderMode.Full


@dataclass
class Camera:
    """lib/core/camera.rs:27-34; defaults scene.rs:45-51"""
    lookfrom: np.ndarray = field(default_factory=lambda: _v3([0.0, 0.0, 0.0]))
    lookat: np.ndarray = field(default_factory=lambda: _v3([0.0, 0.0, -1.0]))
    vertical: np.ndarray = field(default_factory=lambda: _v3([0.0, 1.0, 0.0]))
    vertical_fov: float = 90.0
    aspect_ratio: float = 1.0

    def __post_init__(self):
        self.lookfrom, self.lookat, self.vertical = _v3(self.lookfrom), _v3(self.lookat), _v3(self.vertical)


# --- materials (lib/materials.rs) ---------------------------------------------------------------------
MAT_LAMBERTIAN, MAT_METAL, MAT_EMISSION, MAT_DIELECTRIC, MAT_MIX, MAT_ISOTROPIC = range(6)


class Material:
    kind = -1


@dataclass
class Lambertian(Material):
    albedo: np.ndarray
    kind = MAT_LAMBERTIAN

    def __post_init__(self):
        self.albedo = _v3(self.albedo)


@dataclass
class Metal(Material):
    albedo: np.ndarray
    fuzz: float
    kind = MAT_METAL

    def __post_init__(self):
        self.albedo = _v3(self.albedo)


class Emission(Material):
    """materials.rs:73-91: the constructor pre-multiplies colour by strength in f32; the JSON stores the product."""
    kind = MAT_EMISSION

    def __init__(self, color, strength: float = 1.0):
        self.color = (_v3(color) * f32(strength)).astype(np.float32)


@dataclass
class Dielectric(Material):
    ir: float
    kind = MAT_DIELECTRIC


@dataclass
class MixMaterial(Material):
    """materials.rs:141-157: xi >= factor picks `first`, so factor is the probability of `second`."""
    first: Material
    second: Material
    factor: float
    kind = MAT_MIX


@dataclass
class Isotropic(Material):
    color: np.ndarray
    kind = MAT_ISOTROPIC

    def __post_init__(self):
        self.color = _v3(self.color)


# --- objects ------------------------------------------------------------------------------------------
@dataclass(frozen=True)
class MaterialHandle:
    """scene.rs:67-68"""
    index: int


@dataclass(frozen=True)
class ObjectHandle:
    """scene.rs:70-71"""
    index: int


def _mh(m: Union[int, MaterialHandle]) -> int:
    return m.index if isinstance(m, MaterialHandle) else int(m)


class Sphere:
    """lib/objects.rs:15-20"""

    def __init__(self, center, radius: float, material: Union[int, MaterialHandle]):
        self.center = _v3(center)
        self.radius = float(f32(radius))
        self.material = _mh(material)


class Volume:
    """lib/volume.rs:19-32: a constant-density medium inside a boundary object (`Volume::new(boundary, density)`
    stores -1/density).  The boundary is any Object (`boundary: Box<dyn Object>`): a Sphere, a Mesh, or another Volume (the
    GPU path lowers one level of Volume-in-Volume); its material (normally Isotropic) is the medium's."""

    def __init__(self, boundary: Union["Sphere", "Mesh", "Volume"], density: float):
        if not isinstance(boundary, (Sphere, Mesh, Volume)):
            raise TypeError("Volume boundaries are Sphere, Mesh or Volume")
        self.boundary = boundary
        self.density = float(f32(density))

    @property
    def neg_inv_density(self) -> float:
        stored = getattr(self, "_neg_inv_density", None)
        return stored if stored is not None else float(f32(-1.0) / f32(self.density))  # volume.rs:29

    @property
    def material(self) -> int:
        return self.boundary.material  # Object::material = boundary.material(), volume.rs:98-100


def load_obj(path: str) -> Tuple[np.ndarray, np.ndarray, np.ndarray]:
    """The subset of obj-rs 0.7 `load_obj::<obj::Vertex, u32>` the reference relies on (mesh.rs:63-66):
    triangulated faces with normals (`v//vn` or `v/vt/vn`); one output vertex per distinct
    (position index, normal index) pair in first-appearance order; indices in file face order; groups,
    objects and materials merged; .mtl ignored.  Returns (positions[n,3], normals[n,3], indices[m*3])."""
    pos: List[Tuple[float, float, float]] = []
    nrm: List[Tuple[float, float, float]] = []
    seen = {}
    out_p: List[Tuple[float, float, float]] = []
    out_n: List[Tuple[float, float, float]] = []
    idx: List[int] = []
    with open(path, "r") as fh:
        for line in fh:
            if line.startswith("v "):
                p = line.split()
                pos.append((float(p[1]), float(p[2]), float(p[3])))
            elif line.startswith("vn "):
                p = line.split()
                nrm.append((float(p[1]), float(p[2]), float(p[3])))
            elif line.startswith("f "):
                p = line.split()[1:]
                if len(p) != 3:
                    raise ValueError(f"{path}: only triangulated faces are supported (obj-rs errors out too)")
                for tok in p:
                    parts = tok.split("/")
                    if len(parts) < 3 or parts[2] == "":
                        raise ValueError(f"{path}: face vertex without a normal (obj-rs errors out too)")
                    pi, ni = int(parts[0]), int(parts[2])
                    pi = pi - 1 if pi > 0 else len(pos) + pi
                    ni = ni - 1 if ni > 0 else len(nrm) + ni
                    key = (pi, ni)
                    j = seen.get(key)
                    if j is None:
                        j = len(out_p)
                        seen[key] = j
                        out_p.append(pos[pi])
                        out_n.append(nrm[ni])
                    idx.append(j)
    return (np.asarray(out_p, dtype=np.float64).astype(np.float32).reshape(-1, 3),
            np.asarray(out_n, dtype=np.float64).astype(np.float32).reshape(-1, 3),
            np.asarray(idx, dtype=np.uint32))


class Mesh:
    """lib/core/mesh.rs:31-117.  The per-mesh BVH (Mesh::build_bvh, mesh.rs:119-121) is built by the HIP
    library at upload; `build_bvh()` is kept as a no-op so reference-style construction code reads the same."""

    def __init__(self, positions, normals, indices, material: Union[int, MaterialHandle]):
        self.positions = np.ascontiguousarray(np.asarray(positions, dtype=np.float32).reshape(-1, 3))
        self.normals = np.ascontiguousarray(np.asarray(normals, dtype=np.float32).reshape(-1, 3))
        self.indices = np.ascontiguousarray(np.asarray(indices, dtype=np.uint32).reshape(-1))
        if self.positions.shape != self.normals.shape:
            raise ValueError("positions and normals must have the same shape")
        if self.indices.size % 3:
            raise ValueError("indices must be a multiple of 3 (chunks_exact(3), mesh.rs:85)")
        self.material = _mh(material)

    @classmethod
    def from_file(cls, path: str, material: Union[int, MaterialHandle]) -> "Mesh":
        p, n, i = load_obj(path)
        return cls(p, n, i, material)

    @classmethod
    def from_buffers(cls, positions, normals, indices, material) -> "Mesh":
        return cls(positions, normals, indices, material)

    def build_bvh(self) -> None:
        return None

    @property
    def n_tris(self) -> int:
        return self.indices.size // 3


# --- backgrounds (lib/backgrounds.rs) -------------------------------------------------------------------
@dataclass
class UniformBackground:
    color: np.ndarray

    def __post_init__(self):
        self.color = _v3(self.color)


@dataclass
class GradientBackground:
    top: np.ndarray
    bottom: np.ndarray

    def __post_init__(self):
        self.top, self.bottom = _v3(self.top), _v3(self.bottom)


def read_pfm(path: str) -> np.ndarray:
    """Colour Portable Float Map ("PF"): header `PF\n<width> <height>\n<scale>\n`, then rows of RGB f32 stored bottom
    row first; a negative scale means little-endian samples.  Returns [height, width, 3] float32 with row 0 = top."""
    with open(path, "rb") as fh:
        data = fh.read()
    pos = 0
    tokens = []
    while len(tokens) < 4:  # whitespace-separated header tokens; exactly one whitespace byte ends the header
        while data[pos:pos + 1].isspace():
            pos += 1
        end = pos
        while end < len(data) and not data[end:end + 1].isspace():
            end += 1
        if end == pos:
            raise ValueError("truncated PFM header")
        tokens.append(data[pos:end])
        pos = end
    pos += 1
    if tokens[0] != b"PF":
        raise ValueError("not a colour PFM file (magic must be 'PF'; greyscale 'Pf' is not a sky map)")
    width, height, scale = int(tokens[1]), int(tokens[2]), float(tokens[3])
    if width <= 0 or height <= 0 or scale == 0.0:
        raise ValueError("bad PFM header")
    need = width * height * 3 * 4
    if len(data) - pos < need:
        raise ValueError("truncated PFM pixel data")
    dt = np.dtype("<f4") if scale < 0 else np.dtype(">f4")
    img = np.frombuffer(data, dtype=dt, count=width * height * 3, offset=pos).reshape(height, width, 3)
    return np.ascontiguousarray(img[::-1].astype(np.float32))


def write_pfm(path: str, image) -> None:
    """Little-endian colour PFM of a [height, width, 3] array whose row 0 is the top scanline (tests, tools)."""
    img = np.asarray(image, dtype="<f4")
    with open(path, "wb") as fh:
        fh.write(b"PF\n%d %d\n-1.0\n" % (img.shape[1], img.shape[0]))
        fh.write(np.ascontiguousarray(img[::-1]).tobytes())


class SkyMap:
    """backgrounds.rs:51-92: `image` is height x width x 3 float32, row 0 = top scanline.  The reference
    fills it from an EXR (SkyMap::new); EXR decoding is outside the hot path, so this takes the texels."""

    def __init__(self, image):
        img = np.ascontiguousarray(np.asarray(image, dtype=np.float32))
        if img.ndim != 3 or img.shape[2] != 3 or img.shape[0] == 0 or img.shape[1] == 0:
            raise ValueError("SkyMap image must be HxWx3 and non-empty")
        self.image = img

    @classmethod
    def new(cls, path: str) -> "SkyMap":
        """SkyMap::new (backgrounds.rs:58-81): texels from an OpenEXR file (raytracingrust_amd/exr.py: scanline files,
        none / RLE / ZIPS / ZIP compression), a colour Portable Float Map (`.pfm`) or a float32 `.npy` array of shape
        [height, width, 3]."""
        if path.lower().endswith(".npy"):
            return cls(np.load(path, allow_pickle=False))
        if path.lower().endswith(".pfm"):
            return cls(read_pfm(path))
        from .exr import read_exr

        return cls(read_exr(path))

    @property
    def width(self) -> int:
        return self.image.shape[1]

    @property
    def height(self) -> int:
        return self.image.shape[0]


Background = Union[UniformBackground, GradientBackground, SkyMap]


@dataclass
class LoweredMaterial:
    kind: int
    rgb: np.ndarray
    param: float
    first: int = 0
    second: int = 0


class Scene:
    """lib/core/scene.rs:32-115"""

    def __init__(self, settings: Optional[RenderSettings] = None, camera: Optional[Camera] = None,
                 background: Optional[Background] = None):
        self.camera = camera if camera is not None else Camera()
        self.settings = settings if settings is not None else RenderSettings()
        self.background = background if background is not None else UniformBackground([0.8, 0.8, 0.8])
        self.objects: List[Union[Sphere, Mesh, Volume]] = []
        self.materials: List[Material] = []

    @classmethod
    def default(cls) -> "Scene":
        return cls()

    def add_object(self, obj) -> ObjectHandle:
        self.objects.append(obj)
        return ObjectHandle(len(self.objects) - 1)

    def add_material(self, material: Material) -> MaterialHandle:
        self.materials.append(material)
        return MaterialHandle(len(self.materials) - 1)

    def object(self, h: ObjectHandle):
        return self.objects[h.index]

    def material(self, h: MaterialHandle) -> Material:
        return self.materials[h.index]

    # -- scenes/*.json (serde + typetag) ------------------------------------------
    @classmethod
    def from_file(cls, path: str) -> "Scene":
        """Scene::from_file (scene.rs:74-76).  `settings.clamp_indirect` is absent from both shipped scene
        files, which makes the reference's own loader fail; it defaults to
        Scene::default()'s 10.0 (scene.rs:55) here."""
        with open(path, "r") as fh:
            doc = json.load(fh)
        return cls.from_dict(doc)

    @classmethod
    def from_dict(cls, doc: dict) -> "Scene":
        cam = doc["camera"]
        camera = Camera(_xyz(cam["lookfrom"]), _xyz(cam["lookat"]), _xyz(cam["vertical"]),
                        float(f32(cam["vertical_fov"])), float(f32(cam["aspect_ratio"])))
        st = doc["settings"]
        settings = RenderSettings(
            samples_per_pixel=int(st["samples_per_pixel"]), max_ray_depth=int(st["max_ray_depth"]),
            clamp_indirect=float(f32(st.get("clamp_indirect", 10.0))),
            enable_multithreading=bool(st["enable_multithreading"]), enable_bvh_tree=bool(st["enable_bvh_tree"]),
            mode=RenderMode[st["mode"]])
        scene = cls(settings, camera, cls._background_from_dict(doc["background"]))
        for m in doc["materials"]:
            scene.add_material(cls._material_from_dict(m))
        for o in doc["objects"]:
            t = o["type"]
            if t == "Sphere":
                scene.add_object(Sphere(_xyz(o["center"]), o["radius"], int(o["material"])))
            elif t == "Mesh":
                raise ValueError("Mesh cannot be deserialized (the reference panics: mesh.rs:51-60)")
            elif t == "Volume":
                def volume_from(d):
                    b = d["boundary"]
                    if b.get("type") == "Sphere":
                        inner = Sphere(_xyz(b["center"]), b["radius"], int(b["material"]))
                    elif b.get("type") == "Volume":
                        inner = volume_from(b)
                    else:
                        raise ValueError("a Volume over a Mesh cannot be deserialized (Mesh panics in the reference: mesh.rs:51-60)")
                    vol = Volume(inner, 1.0)
                    vol.density = float(f32(-1.0) / f32(d["neg_inv_density"]))
                    vol._neg_inv_density = float(f32(d["neg_inv_density"]))  # keep the stored value bit for bit
                    return vol

                scene.add_object(volume_from(o))
            else:
                raise ValueError(f"unknown object type {t!r}")
        return scene

    # -- the writer side of the same layout (the reference derives Serialize: `serde_json::to_string(&scene)`) -----
    def to_dict(self) -> dict:
        """The serde + typetag document of this scene: what `Scene::from_file` reads back.
        f32 fields are written in their shortest round-tripping decimal form, as serde_json (ryu) prints them.
        A Mesh cannot be serialised: the reference's `Serialize for Mesh` panics (mesh.rs:40-49)."""

        def num(x) -> float:
            return float(str(np.float32(x)))  # shortest decimal that parses back to the same f32

        def xyz(v) -> dict:
            return {"x": num(v[0]), "y": num(v[1]), "z": num(v[2])}

        def rgb(v) -> dict:
            return {"r": num(v[0]), "g": num(v[1]), "b": num(v[2])}

        def material(m: Material) -> dict:
            if isinstance(m, Lambertian):
                return {"type": "Lambertian", "albedo": rgb(m.albedo)}
            if isinstance(m, Metal):
                return {"type": "Metal", "albedo": rgb(m.albedo), "fuzz": num(m.fuzz)}
            if isinstance(m, Emission):
                return {"type": "Emission", "color": rgb(m.color)}  # strength is pre-multiplied (materials.rs:79-83)
            if isinstance(m, Dielectric):
                return {"type": "Dielectric", "ir": num(m.ir)}
            if isinstance(m, MixMaterial):
                return {"type": "MixMaterial", "first": material(m.first), "second": material(m.second), "factor": num(m.factor)}
            if isinstance(m, Isotropic):
                return {"type": "Isotropic", "color": rgb(m.color)}
            raise TypeError(f"unknown material {type(m).__name__}")

        def sphere(o: Sphere) -> dict:
            return {"type": "Sphere", "center": xyz(o.center), "radius": num(o.radius), "material": int(o.material)}

        objects = []
        for o in self.objects:
            if isinstance(o, Sphere):
                objects.append(sphere(o))
            elif isinstance(o, Volume):
                def volume_to(v: Volume) -> dict:
                    if isinstance(v.boundary, Sphere):
                        b = sphere(v.boundary)
                    elif isinstance(v.boundary, Volume):
                        b = volume_to(v.boundary)
                    else:
                        raise ValueError("a Volume over a Mesh cannot be serialized (Mesh panics in the reference: mesh.rs:40-49)")
                    return {"type": "Volume", "boundary": b, "neg_inv_density": num(v.neg_inv_density)}

                objects.append(volume_to(o))
            else:
                raise ValueError("Mesh cannot be serialized (the reference panics: mesh.rs:40-49)")
        bg = self.background
        if isinstance(bg, UniformBackground):
            background = {"type": "UniformBackground", "color": rgb(bg.color)}
        elif isinstance(bg, GradientBackground):
            background = {"type": "GradientBackground", "top": rgb(bg.top), "bottom": rgb(bg.bottom)}
        else:  # backgrounds.rs:51-56: the whole image inline
            background = {"type": "SkyMap", "image": [rgb(t) for t in bg.image.reshape(-1, 3)],
                          "width": bg.width, "height": bg.height}
        c, st = self.camera, self.settings
        return {
            "camera": {"lookfrom": xyz(c.lookfrom), "lookat": xyz(c.lookat), "vertical": xyz(c.vertical),
                       "vertical_fov": num(c.vertical_fov), "aspect_ratio": num(c.aspect_ratio)},
            "settings": {"samples_per_pixel": int(st.samples_per_pixel), "max_ray_depth": int(st.max_ray_depth),
                         "clamp_indirect": num(st.clamp_indirect), "enable_multithreading": bool(st.enable_multithreading),
                         "enable_bvh_tree": bool(st.enable_bvh_tree), "mode": RenderMode(st.mode).name},
            "background": background,
            "objects": objects,
            "materials": [material(m) for m in self.materials],
        }

    def to_file(self, path: str) -> None:
        with open(path, "w") as fh:
            json.dump(self.to_dict(), fh, indent=2)

    @staticmethod
    def _background_from_dict(b: dict) -> Background:
        t = b["type"]
        if t == "UniformBackground":
            return UniformBackground(_rgb(b["color"]))
        if t == "GradientBackground":
            return GradientBackground(_rgb(b["top"]), _rgb(b["bottom"]))
        if t == "SkyMap":
            w, h = int(b["width"]), int(b["height"])
            img = np.asarray([[p["r"], p["g"], p["b"]] for p in b["image"]], dtype=np.float32).reshape(h, w, 3)
            return SkyMap(img)
        raise ValueError(f"unknown background type {t!r}")

    @classmethod
    def _material_from_dict(cls, m: dict) -> Material:
        t = m["type"]
        if t == "Lambertian":
            return Lambertian(_rgb(m["albedo"]))
        if t == "Metal":
            return Metal(_rgb(m["albedo"]), float(f32(m["fuzz"])))
        if t == "Emission":
            return Emission(_rgb(m["color"]), 1.0)
        if t == "Dielectric":
            return Dielectric(float(f32(m["ir"])))
        if t == "MixMaterial":
            return MixMaterial(cls._material_from_dict(m["first"]), cls._material_from_dict(m["second"]),
                               float(f32(m["factor"])))
        if t == "Isotropic":
            return Isotropic(_rgb(m["color"]))
        raise ValueError(f"unknown material type {t!r}")

    # -- lowering to plain-old-data -----------------------------------------------------------------------
    def lowered_materials(self) -> List[LoweredMaterial]:
        """Scene.materials with MixMaterial's nested boxes flattened: the first len(self.materials) entries
        keep their handles; children are appended after them (index > parent, as the ABI requires)."""
        out: List[Optional[LoweredMaterial]] = [None] * len(self.materials)

        def lower(m: Material, slot: Optional[int]) -> int:
            if slot is None:
                out.append(None)
                slot = len(out) - 1
            if isinstance(m, Lambertian):
                out[slot] = LoweredMaterial(MAT_LAMBERTIAN, m.albedo, 0.0)
            elif isinstance(m, Metal):
                out[slot] = LoweredMaterial(MAT_METAL, m.albedo, float(f32(m.fuzz)))
            elif isinstance(m, Emission):
                out[slot] = LoweredMaterial(MAT_EMISSION, m.color, 0.0)
            elif isinstance(m, Dielectric):
                out[slot] = LoweredMaterial(MAT_DIELECTRIC, _v3([0, 0, 0]), float(f32(m.ir)))
            elif isinstance(m, Isotropic):
                out[slot] = LoweredMaterial(MAT_ISOTROPIC, m.color, 0.0)
            elif isinstance(m, MixMaterial):
                out[slot] = LoweredMaterial(MAT_MIX, _v3([0, 0, 0]), float(f32(m.factor)))
                a = lower(m.first, None)
                b = lower(m.second, None)
                out[slot].first, out[slot].second = a, b
            else:
                raise TypeError(f"not a material: {m!r}")
            return slot

        for i, m in enumerate(self.materials):
            lower(m, i)
        return out  # type: ignore[return-value]


class LoweredScene:
    """ctypes rtr_scene_desc plus the numpy buffers it points into (kept alive here)."""

    def __init__(self, scene: Scene):
        self._keep = []
        spheres = [o for o in scene.objects if isinstance(o, Sphere)]
        obj_volumes = [o for o in scene.objects if isinstance(o, Volume)]
        # volumes[]: the scene's Volume objects, then the Volumes that only bound another Volume (not objects themselves), in
        # discovery order; a Volume names its boundary Volume by index (boundary_kind = 2)
        volumes = list(obj_volumes)
        k = 0
        while k < len(volumes):
            if isinstance(volumes[k].boundary, Volume):
                volumes.append(volumes[k].boundary)
            k += 1
        # meshes[]: the scene's Mesh objects, then the meshes that only bound a Volume (not objects themselves)
        meshes = [o for o in scene.objects if isinstance(o, Mesh)] + [v.boundary for v in volumes if isinstance(v.boundary, Mesh)]
        n_obj_meshes = sum(isinstance(o, Mesh) for o in scene.objects)
        if len(spheres) + n_obj_meshes + len(obj_volumes) != len(scene.objects):
            raise TypeError("Scene.objects may hold Sphere, Mesh and Volume only")
        vol = (_capi.rtr_volume * max(len(volumes), 1))()
        vi = 0
        n_obj = len(scene.objects)
        objs = (_capi.rtr_object * max(n_obj, 1))()
        sph = (_capi.rtr_sphere * max(len(spheres), 1))()
        msh = (_capi.rtr_mesh * max(len(meshes), 1))()
        for mi, o in enumerate(meshes):
            self._keep += [o.positions, o.normals, o.indices]
            msh[mi].positions = o.positions.ctypes.data_as(C.POINTER(C.c_float))
            msh[mi].normals = o.normals.ctypes.data_as(C.POINTER(C.c_float))
            msh[mi].indices = o.indices.ctypes.data_as(C.POINTER(C.c_uint32))
            msh[mi].n_verts = o.positions.shape[0]
            msh[mi].n_tris = o.n_tris
            msh[mi].material = o.material
        si = mi = 0
        bi = n_obj_meshes  # next boundary-only mesh
        for i, o in enumerate(scene.objects):
            if isinstance(o, Sphere):
                objs[i].kind, objs[i].index = 0, si
                sph[si].center = _capi.c_float3(*[float(x) for x in o.center])
                sph[si].radius = o.radius
                sph[si].material = o.material
                si += 1
            elif isinstance(o, Volume):
                objs[i].kind, objs[i].index = 2, vi
                vi += 1
            else:
                objs[i].kind, objs[i].index = 1, mi
                mi += 1
        for vi, o in enumerate(volumes):  # (objects first, boundary-only Volumes behind them)
            b = o.boundary
            if isinstance(b, Sphere):
                vol[vi].boundary.center = _capi.c_float3(*[float(x) for x in b.center])
                vol[vi].boundary.radius = b.radius
                vol[vi].boundary.material = b.material
                vol[vi].boundary_kind = 0
            elif isinstance(b, Mesh):
                vol[vi].boundary.material = b.material
                vol[vi].boundary_kind, vol[vi].boundary_mesh = 1, bi
                bi += 1
            else:  # a Volume: named by its index in volumes[]
                vol[vi].boundary_kind, vol[vi].boundary_mesh = 2, next(j for j, w in enumerate(volumes) if w is b)
            vol[vi].neg_inv_density = o.neg_inv_density
        lm = scene.lowered_materials()
        mats = (_capi.rtr_material * max(len(lm), 1))()
        for i, m in enumerate(lm):
            mats[i].kind = m.kind
            mats[i].rgb = _capi.c_float3(*[float(x) for x in m.rgb])
            mats[i].param = m.param
            mats[i].first, mats[i].second = m.first, m.second
        d = _capi.rtr_scene_desc()
        d.objects, d.n_objects = objs, n_obj
        d.spheres, d.n_spheres = sph, len(spheres)
        d.meshes, d.n_meshes = msh, len(meshes)
        d.materials, d.n_materials = mats, len(lm)
        d.volumes, d.n_volumes = vol, len(volumes)
        bg = scene.background
        if isinstance(bg, UniformBackground):
            d.background.kind = 0
            d.background.a = _capi.c_float3(*[float(x) for x in bg.color])
        elif isinstance(bg, GradientBackground):
            d.background.kind = 1
            d.background.a = _capi.c_float3(*[float(x) for x in bg.top])
            d.background.b = _capi.c_float3(*[float(x) for x in bg.bottom])
        elif isinstance(bg, SkyMap):
            d.background.kind = 2
            self._keep.append(bg.image)
            d.background.image = bg.image.ctypes.data_as(C.POINTER(C.c_float))
            d.background.width, d.background.height = bg.width, bg.height
        else:
            raise TypeError(f"not a background: {bg!r}")
        self._keep += [objs, sph, msh, mats, vol]
        self.desc = d


def camera_struct(cam: Camera) -> _capi.rtr_camera:
    c = _capi.rtr_camera()
    c.lookfrom = _capi.c_float3(*[float(x) for x in cam.lookfrom])
    c.lookat = _capi.c_float3(*[float(x) for x in cam.lookat])
    c.vertical = _capi.c_float3(*[float(x) for x in cam.vertical])
    c.vertical_fov = float(f32(cam.vertical_fov))
    c.aspect_ratio = float(f32(cam.aspect_ratio))
    return c


def settings_struct(st: RenderSettings) -> _capi.rtr_settings:
    s = _capi.rtr_settings()
    s.samples_per_pixel = int(st.samples_per_pixel)
    s.max_ray_depth = int(st.max_ray_depth)
    s.clamp_indirect = float(f32(st.clamp_indirect))
    s.enable_multithreading = 1 if st.enable_multithreading else 0
    s.enable_bvh_tree = 1 if st.enable_bvh_tree else 0
    s.mode = int(st.mode)
    return s
