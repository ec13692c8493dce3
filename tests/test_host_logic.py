"""CPU-side tests of the product's host logic (no GPU, no compute calls): the C ABI exports every symbol
include/rtr_hip.h declares, host-only contexts refuse to render, the library's BVH builder produces the
oracle's (= the reference's) tree, and the scene / OBJ readers and the POD lowering behave."""
import ctypes as C
import json
import os
import re

import numpy as np
import pytest

import pyoracle
import raytracingrust_amd as rt
from raytracingrust_amd import _capi
from helpers import ROOT, benchmark_scene, house_scene, mesh_path, monkey_scene, scene_path, simple_scene


def test_library_exports_every_declared_symbol():
    header = open(os.path.join(ROOT, "include", "rtr_hip.h")).read()
    declared = set(re.findall(r"\b(rtr_[a-z0-9_]+)\s*\(", header))
    assert len(declared) >= 16
    lib = C.CDLL(_capi.library_path())
    for name in sorted(declared):
        assert hasattr(lib, name), f"librtr_hip.so does not export {name}"
    bound = {n for n, _, _ in _capi.SYMBOLS}
    assert declared == bound, declared ^ bound
    assert _capi.load().rtr_abi_version() == 5


@pytest.mark.parametrize("case", __import__("kat_cases").HOST_CASES, ids=lambda c: c.__name__)
def test_host_known_answers_on_the_product(case):
    """Camera::ray_origin and the BVH builder's tree shape, asserted on librtr_hip.so's host code (no device needed)."""
    import kat_cases

    b = kat_cases.HipBackend(-1)
    try:
        case(b)
    finally:
        b.close()


def test_struct_layouts_match_the_header():
    assert C.sizeof(_capi.rtr_sphere) == 20 and C.sizeof(_capi.rtr_material) == 28
    assert C.sizeof(_capi.rtr_object) == 8 and C.sizeof(_capi.rtr_camera) == 44
    assert C.sizeof(_capi.rtr_settings) == 24 and C.sizeof(_capi.rtr_mesh) == 40
    assert C.sizeof(_capi.rtr_background) == 48 and C.sizeof(_capi.rtr_scene_desc) == 64 + 48 + 16
    assert C.sizeof(_capi.rtr_volume) == 32  # ABI 4: + boundary_kind, boundary_mesh
    assert C.sizeof(_capi.rtr_hit) == 48


def test_host_only_context_refuses_to_render():
    """There is no CPU fallback: without a device every compute entry point is an error."""
    with rt.Renderer(-1) as r:
        r.upload_scene(benchmark_scene(1))
        with pytest.raises(rt.RtrError) as e:
            r.render(8, 8)
        assert e.value.code == -2  # RTR_ERR_NO_DEVICE
        with pytest.raises(rt.RtrError):
            r.render_frame_device(8, 8, 1, 0x1000)
        with pytest.raises(rt.RtrError):
            r.resolve_device(0x1000, 1, 8, 8, 0x2000)


def test_render_before_upload_and_bad_arguments():
    with rt.Renderer(-1) as r:
        with pytest.raises(rt.RtrError):
            r.scene_info()
        s = benchmark_scene(1)
        s.objects[0].material = 99  # out of bounds: scene.rs:106-110 would panic
        with pytest.raises(rt.RtrError) as e:
            r.upload_scene(s)
        assert e.value.code == -4
        s = rt.Scene()
        m = s.add_material(rt.Lambertian([0.5, 0.5, 0.5]))
        # NaN on z: every spread comparison is false, the axis falls through to z and the sort key is NaN
        s.add_object(rt.Sphere([0, 0, float("nan")], 1.0, m))
        s.add_object(rt.Sphere([0, 0, 0], 1.0, m))
        with pytest.raises(rt.RtrError) as e:
            r.upload_scene(s)
        assert e.value.code == -5  # NaN centroid: partial_cmp().unwrap() panics in the reference
        with pytest.raises(pyoracle.OracleError):
            pyoracle.OracleScene(s).render(4, 4, 1)
        s = rt.Scene()
        s.add_material(rt.Lambertian([0.5, 0.5, 0.5]))
        s.add_object(rt.Mesh(np.zeros((3, 3), np.float32), np.zeros((3, 3), np.float32), [0, 1, 7], 0))
        with pytest.raises(rt.RtrError) as e:
            r.upload_scene(s)
        assert e.value.code == -4


def _leaf_pairs_from_oracle(scene):
    o = pyoracle.OracleScene(scene)
    out = []
    for obj in o.bvh_leaf_order():
        if isinstance(scene.objects[int(obj)], rt.Mesh):
            out += [(int(obj), int(t)) for t in o.mesh_leaf_order(int(obj))]
        else:
            out.append((int(obj), 0))
    return out


@pytest.mark.parametrize("builder", [lambda: simple_scene(1), lambda: benchmark_scene(1), lambda: monkey_scene(1),
                                     lambda: house_scene(1)])
def test_bvh_builder_matches_oracle_leaf_order(builder):
    """The library's builder (csrc/rtr_scene.cpp) and the oracle's (oracle/rt_oracle.c) are independent
    implementations of bvh.rs:59-144; their depth-first leaf orders over the two-level tree must agree."""
    scene = builder()
    with rt.Renderer(-1) as r:
        r.upload_scene(scene)
        got = [tuple(int(v) for v in row) for row in r.bvh_leaf_order()]
        info = r.scene_info()
    want = _leaf_pairs_from_oracle(scene)
    assert got == want
    n_tris = sum(o.n_tris for o in scene.objects if isinstance(o, rt.Mesh))
    n_sph = sum(1 for o in scene.objects if isinstance(o, rt.Sphere))
    assert info.n_leaves == n_tris + n_sph and info.n_triangles == n_tris
    assert info.n_nodes == info.n_leaves - 1  # every Split has two non-empty children (bvh.rs:126-135)


def test_bvh_builder_random_spheres_and_stress_shapes():
    """benches/bvh_tree_stress.rs geometry (a 20^3 lattice would be 8000; 9^3 here) + random radii."""
    rng = np.random.default_rng(11)
    s = rt.Scene()
    m = s.add_material(rt.Dielectric(1.5))
    for x in range(-4, 5):
        for y in range(-4, 5):
            for z in range(-4, 5):
                s.add_object(rt.Sphere([x, y, z], 0.5, m))
    for _ in range(300):
        s.add_object(rt.Sphere(rng.uniform(-5, 5, size=3), float(rng.uniform(0.01, 2.0)), m))
    with rt.Renderer(-1) as r:
        r.upload_scene(s)
        got = [int(v) for v in r.bvh_leaf_order()[:, 0]]
        info = r.scene_info()
    assert got == [int(v) for v in pyoracle.OracleScene(s).bvh_leaf_order()]
    assert sorted(got) == list(range(len(s.objects)))
    assert info.tree_height == int(np.ceil(np.log2(len(s.objects))))


def test_degenerate_scenes():
    with rt.Renderer(-1) as r:
        s = rt.Scene()
        r.upload_scene(s)  # empty: BvhNode::None
        assert r.scene_info().n_leaves == 0 and r.scene_info().n_nodes == 0
        m = s.add_material(rt.Lambertian([0.5, 0.5, 0.5]))
        s.add_object(rt.Sphere([0, 0, -1], 0.5, m))
        r.upload_scene(s)  # one object: a bare leaf, no box at all (bvh.rs:62)
        assert r.scene_info().n_leaves == 1 and r.scene_info().n_nodes == 0 and r.scene_info().tree_height == 0
        s.add_object(rt.Mesh(np.zeros((0, 3), np.float32), np.zeros((0, 3), np.float32), [], m))  # empty mesh
        r.upload_scene(s)
        assert r.scene_info().n_leaves == 1 and r.scene_info().n_nodes == 1


def test_scene_json_reader():
    s = rt.Scene.from_file(scene_path("benchmark.json"))
    assert len(s.objects) == 5 and len(s.materials) == 5
    assert s.settings.samples_per_pixel == 5 and s.settings.max_ray_depth == 6 and not s.settings.enable_bvh_tree
    assert s.settings.clamp_indirect == 10.0  # absent from the file: Scene::default()'s value (scene.rs:55)
    assert s.settings.mode == rt.RenderMode.Full and s.camera.aspect_ratio == np.float32(1.2)
    assert isinstance(s.background, rt.UniformBackground) and np.allclose(s.background.color, 0.6)
    kinds = [type(m).__name__ for m in s.materials]
    assert kinds == ["Lambertian", "Lambertian", "Metal", "Dielectric", "Emission"]
    assert np.array_equal(s.materials[4].color, np.float32([2, 2, 2]))
    assert s.objects[0].material == 4 and s.objects[0].radius == 10.0
    s2 = rt.Scene.from_file(scene_path("simple.json"))
    assert len(s2.objects) == 102 and len(s2.materials) == 102 and s2.settings.enable_bvh_tree
    counts = {}
    for m in s2.materials:
        counts[type(m).__name__] = counts.get(type(m).__name__, 0) + 1
    assert counts == {"Lambertian": 32, "Metal": 36, "Dielectric": 18, "Emission": 16}
    with pytest.raises(ValueError):
        rt.Scene.from_dict({"camera": {"lookfrom": {"x": 0, "y": 0, "z": 0}, "lookat": {"x": 0, "y": 0, "z": -1},
                                       "vertical": {"x": 0, "y": 1, "z": 0}, "vertical_fov": 90, "aspect_ratio": 1},
                            "settings": {"samples_per_pixel": 1, "max_ray_depth": 1, "enable_multithreading": True,
                                         "enable_bvh_tree": True, "mode": "Full"},
                            "background": {"type": "UniformBackground", "color": {"r": 0, "g": 0, "b": 0}},
                            "materials": [], "objects": [{"type": "Mesh"}]})


def test_obj_reader_matches_obj_rs_counts():
    """one vertex per distinct (position, normal) index pair, first-appearance order."""
    for name, tris, verts in [("monkey.obj", 967, 2865), ("house.obj", 9153, 9919), ("plane.obj", 2, 4),
                              ("cube.obj", 12, None), ("cylinder.obj", 124, None)]:
        p, n, i = rt.load_obj(mesh_path(name))
        assert i.size == 3 * tris and p.shape == n.shape
        if verts is not None:
            assert p.shape[0] == verts
        assert int(i.max()) == p.shape[0] - 1
        first_use = [np.flatnonzero(i == v)[0] for v in range(min(p.shape[0], 50))]
        assert first_use == sorted(first_use)
    p, n, i = rt.load_obj(mesh_path("plane.obj"))
    assert np.array_equal(p[0], np.float32([10, 0, 10])) and np.all(n == np.float32([0, 1, 0]))
    assert list(i) == [0, 1, 2, 0, 3, 1]


def test_material_lowering_flattens_mix_children_after_parents():
    s = rt.Scene()
    s.add_material(rt.Lambertian([0.1, 0.2, 0.3]))
    s.add_material(rt.MixMaterial(rt.Metal([0.1, 0.1, 0.1], 0.0),
                                  rt.MixMaterial(rt.Lambertian([0.4, 0.4, 0.4]), rt.Dielectric(1.5), 0.25), 0.5))
    s.add_material(rt.Emission([1.0, 1.0, 1.0], 50.0))
    lm = s.lowered_materials()
    assert [m.kind for m in lm[:3]] == [rt.scene.MAT_LAMBERTIAN, rt.scene.MAT_MIX, rt.scene.MAT_EMISSION]
    assert len(lm) == 7
    mix = lm[1]
    assert mix.first > 1 and mix.second > 1 and lm[mix.first].kind == rt.scene.MAT_METAL
    inner = lm[mix.second]
    assert inner.kind == rt.scene.MAT_MIX and inner.first > mix.second and inner.second > mix.second
    assert np.array_equal(lm[2].rgb, np.float32([50, 50, 50]))
    with rt.Renderer(-1) as r:
        s.add_object(rt.Sphere([0, 0, -1], 0.5, 1))
        r.upload_scene(s)
        assert r.scene_info().n_materials == 7


def test_render_target_and_defaults_mirror_the_reference():
    t = rt.RenderTarget(7, 5)
    assert t.data.shape == (7 * 5 * 4,) and t.data.dtype == np.uint8 and not t.request_redraw
    s = rt.Scene.default()  # scene.rs:42-65
    assert s.settings.samples_per_pixel == 5 and s.settings.max_ray_depth == 6 and s.settings.clamp_indirect == 10.0
    assert s.settings.enable_bvh_tree and s.settings.enable_multithreading and s.settings.mode == rt.RenderMode.Full
    assert np.array_equal(s.camera.lookat, np.float32([0, 0, -1])) and s.camera.vertical_fov == 90.0
    assert np.allclose(s.background.color, 0.8)
    h = s.add_material(rt.Dielectric(1.5))
    assert h == rt.MaterialHandle(0) and s.add_object(rt.Sphere([0, 0, 0], 1, h)) == rt.ObjectHandle(0)


@pytest.mark.parametrize("name", ["benchmark.json", "simple.json"])
def test_scene_json_round_trip(name, tmp_path):
    """the serde + typetag document written back equals the shipped file (plus `clamp_indirect`, which
    both files lack and the reference's own loader therefore rejects) and reads back to the same scene."""
    import json

    from raytracingrust_amd.configs import scene_path

    with open(scene_path(name)) as fh:
        shipped = json.load(fh)
    scene = rt.Scene.from_file(scene_path(name))
    doc = scene.to_dict()
    assert doc["settings"].pop("clamp_indirect") == 10.0
    assert doc == shipped
    out = str(tmp_path / name)
    scene.to_file(out)
    again = rt.Scene.from_file(out)
    assert again.to_dict() == scene.to_dict()
    assert len(scene.objects) == len(again.objects) and len(scene.materials) == len(again.materials)


def test_scene_json_writer_covers_every_type_and_refuses_meshes(tmp_path):
    s = rt.Scene()
    s.background = rt.GradientBackground([1.0, 1.0, 1.0], [0.5, 0.7, 1.0])
    lam = s.add_material(rt.Lambertian([0.1, 0.2, 0.3]))
    s.add_material(rt.Metal([0.9, 0.8, 0.7], 0.03))
    s.add_material(rt.Emission([1.0, 0.5, 0.25], 4.0))
    s.add_material(rt.Dielectric(1.5))
    iso = s.add_material(rt.Isotropic([0.8, 0.8, 0.9]))
    s.add_material(rt.MixMaterial(rt.Metal([0.1, 0.1, 0.1], 0.0), rt.MixMaterial(rt.Lambertian([0.4, 0.4, 0.4]), rt.Dielectric(1.3), 0.25), 0.5))
    s.add_object(rt.Sphere([0.0, -100.5, -1.0], 100.0, lam))
    s.add_object(rt.Volume(rt.Sphere([0.0, 0.0, 0.0], 3.0, iso), 0.7))
    s.settings.mode = rt.RenderMode.Clay
    p = str(tmp_path / "all.json")
    s.to_file(p)
    back = rt.Scene.from_file(p)
    assert back.to_dict() == s.to_dict()
    assert back.objects[1].neg_inv_density == s.objects[1].neg_inv_density  # bit for bit through the decimal form
    assert np.array_equal(back.materials[2].color, s.materials[2].color)    # emission: strength stays multiplied in
    sky = rt.Scene()
    sky.background = rt.SkyMap(np.arange(24, dtype=np.float32).reshape(2, 4, 3) / 7.0)
    sky.to_file(p)
    assert np.array_equal(rt.Scene.from_file(p).background.image, sky.background.image)
    s.add_object(rt.Mesh([[0, 0, 0], [1, 0, 0], [0, 1, 0]], [[0, 0, 1]] * 3, [0, 1, 2], lam))
    with pytest.raises(ValueError):
        s.to_dict()


def test_options_are_validated_without_a_device():
    """rtr_set_option works on a host-only context: known names take their documented values, anything else is
    RTR_ERR_INVALID_ARGUMENT; tree options change the flattened tree, kernel options do not."""
    with rt.Renderer(-1) as r:
        for name, good, bad in [("mesh_tree", ["reference", "sah", "lbvh", "reference"], "octree"), ("scene_tree", ["sah", "reference"], ""),
                                ("leaf_pairs", ["off", "on"], "maybe"), ("tree_kernel", ["wave", "resumable", "samples", "stream", "auto"], "fast"),
                                ("stream_flush", ["1", "64", "8"], "0"),
                                ("samples_batch", ["3", "0"], "-1"),
                                ("sample_chunks", ["1", "7", "auto"], "0"),
                                ("debug_chunk_taper", ["0", "7", "auto"], "8"), ("debug_chunk_lengths", ["24:7:1", "5"], "3::1"),
                                ("chunk_timeout_ms", ["0", "2.5", "10000"], "-1"), ("debug_fault", ["on", "off"], "1"),
                                ("debug_alloc_limit", ["4096", "0"], "lots"),
                                ("sky_filter", ["bilinear", "nearest"], "cubic"),
                                ("stack_entries", ["wide", "auto"], "narrow"), ("block_samples", ["1", "64", "8"], "0"), ("debug_lds_pad", ["64", "0"], "many")]:
            for v in good:
                r.set_option(name, v)
            with pytest.raises(rt.RtrError):
                r.set_option(name, bad)
        with pytest.raises(rt.RtrError):
            r.set_option("no_such_option", "1")
        # a host-only context has no device fault word to report: the synchronising readers stay RTR_OK / empty
        assert r.stats().launches == 0 and len(r.kernel_times_ms()) == 0 and len(r.integrate_times_ms()) == 0
        assert len(r.segment_counts()) == 0 and r.kernel_name() == ""
        s = simple_scene(1)
        r.upload_scene(s)
        base = r.scene_info()
        r.set_option("leaf_pairs", "off")
        r.upload_scene(s)
        off = r.scene_info()
        assert (off.n_nodes, off.n_leaves, off.tree_height) == (base.n_nodes, base.n_leaves, base.tree_height)
        assert off.device_bytes >= base.device_bytes  # unfolded pairs are node records again


def test_render_target_writes_png_and_ppm(tmp_path):
    from helpers import read_png, read_ppm

    t = rt.RenderTarget(37, 11)
    t.data[:] = (np.arange(t.data.size) * 13 % 251).astype(np.uint8)
    t.save(str(tmp_path / "a.png"))
    t.save(str(tmp_path / "a.ppm"))
    np.testing.assert_array_equal(read_png(str(tmp_path / "a.png")), t.image())
    np.testing.assert_array_equal(read_ppm(str(tmp_path / "a.ppm")), t.image()[:, :, :3])


def test_multi_gpu_sample_split_is_a_partition():
    """rtr_create_multi's sample split (the library's own arithmetic, rtr_sample_range): the ranks' ranges are contiguous,
    disjoint and cover the frame's samples; equal to the Python driver's (raytracingrust_amd/distributed.py)."""
    from raytracingrust_amd.distributed import sample_range

    L = _capi.load()
    for n in (1, 2, 3, 7, 8, 64):
        for spp in (0, 1, 5, 63, 256, 257, 1024):
            end = 0
            for r in range(n):
                b, c = C.c_uint32(), C.c_uint32()
                assert L.rtr_sample_range(r, n, spp, C.byref(b), C.byref(c)) == 0
                assert (b.value, c.value) == tuple(sample_range(r, n, spp))
                assert b.value == end and c.value in (spp // n, spp // n + 1)
                end += c.value
            assert end == spp
    b, c = C.c_uint32(), C.c_uint32()
    assert L.rtr_sample_range(2, 2, 8, C.byref(b), C.byref(c)) == -1  # rank out of range
    assert L.rtr_sample_range(0, 0, 8, C.byref(b), C.byref(c)) == -1


def test_host_only_multi_context():
    """A context over several (host-only) ranks lowers and flattens the scene once and answers the introspection calls;
    every compute call fails with RTR_ERR_NO_DEVICE; mixed or repeated-and-mixed device lists are rejected."""
    with rt.Renderer([-1, -1, -1]) as r:
        r.upload_scene(simple_scene(2))
        with rt.Renderer(-1) as single:
            single.upload_scene(simple_scene(2))
            assert np.array_equal(r.bvh_leaf_order(), single.bvh_leaf_order())
            assert r.scene_info().n_leaves == single.scene_info().n_leaves == 102
        with pytest.raises(rt.RtrError) as e:
            r.render(8, 8)
        assert e.value.code == -2
        with pytest.raises(rt.RtrError) as e:
            r.render_samples_device(8, 8, 1, 0, 1, 0x1000)
        assert e.value.code == -8  # a multi-GPU context shards its frames itself
        r.set_option("multi_split", "tiles")
        with pytest.raises(rt.RtrError):
            r.set_option("multi_split", "rows")
    with rt.Renderer(-1) as single:
        with pytest.raises(rt.RtrError):
            single.set_option("multi_split", "tiles")
    for bad in ([0, -1], [0, 1, 0], []):
        with pytest.raises(rt.RtrError):
            rt.Renderer(bad)


def test_volume_over_a_volume_lowers_to_one_leaf():
    """lib/volume.rs:20 admits any Object as a Volume's boundary.  One level of Volume-in-Volume lowers to a single leaf (the
    innermost boundary's box and material, volume.rs:90-100) in the library and evaluates recursively in the oracle; deeper
    chains are outside the GPU path; the typetag JSON round-trips the nesting."""
    s = rt.Scene()
    fog = s.add_material(rt.Isotropic([1.0, 1.0, 1.0]))
    s.add_object(rt.Volume(rt.Volume(rt.Sphere([0, 0, -2], 1.0, fog), 2.0), 3.0))
    s.add_object(rt.Sphere([2, 0, -2], 0.5, fog))
    with rt.Renderer(-1) as r:
        r.upload_scene(s)
        info = r.scene_info()
        assert info.n_leaves == 2 and info.n_objects == 2
    again = rt.Scene.from_dict(json.loads(json.dumps(s.to_dict())))
    v = again.objects[0]
    assert isinstance(v, rt.Volume) and isinstance(v.boundary, rt.Volume) and isinstance(v.boundary.boundary, rt.Sphere)
    assert v.neg_inv_density == s.objects[0].neg_inv_density and v.boundary.neg_inv_density == s.objects[0].boundary.neg_inv_density
    _, mean, st = pyoracle.OracleScene(s).render(24, 16, 3)
    _, mean2, st2 = pyoracle.OracleScene(again).render(24, 16, 3)
    assert st.segments == st2.segments and np.array_equal(mean, mean2)
    deep = rt.Scene()
    deep.add_material(rt.Isotropic([1.0, 1.0, 1.0]))
    deep.add_object(rt.Volume(rt.Volume(rt.Volume(rt.Sphere([0, 0, -2], 1.0, 0), 2.0), 3.0), 4.0))
    with rt.Renderer(-1) as r:
        with pytest.raises(rt.RtrError) as e:
            r.upload_scene(deep)
        assert e.value.code == -8  # RTR_ERR_UNSUPPORTED
    assert pyoracle.OracleScene(deep).render(8, 8, 1)[2].segments > 0  # the oracle follows the boxes to any depth
