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
    assert c["mismatched_pixels"] == 0, c  # the culled walk (relative margin, kCullMargin) vs the unculled reference walk


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


@pytest.mark.parametrize("n_shards", [1, 2, 3, 8])
def test_sharded_tiles_plus_resolve_equal_fused_frame(gpu_renderer, n_shards):
    """The multi-GPU decomposition on one GPU: every shard renders its 8x8 tiles into its own zeroed float4
    buffer, the buffers are summed (what the RCCL reduce does), rtr_resolve_device finishes the frame.  Must be
    bit-identical to the fused single-kernel frame (x + 0 = x) and to the oracle."""
    import torch

    from raytracingrust_amd.distributed import owned_pixel_mask

    w, h, seed = 100, 52, 99  # not multiples of 8: ragged edge tiles
    scene = simple_scene(4)
    r = gpu_renderer
    r.upload_scene(scene)
    rgba_f, mean_f = r.render(w, h, seed=seed, want_rgba8=True, want_mean=True)
    seg_f = r.stats().segments
    dev = torch.device("cuda", 0)
    total = torch.zeros((h, w, 3), dtype=torch.float32, device=dev)
    segs = 0
    for shard in range(n_shards):
        part = torch.zeros((h, w, 3), dtype=torch.float32, device=dev)
        torch.cuda.synchronize()
        r.render_tiles_device(w, h, seed, shard, n_shards, part.data_ptr(), 0)
        st = r.stats()  # synchronises
        segs += st.segments
        mask = torch.from_numpy(owned_pixel_mask(w, h, shard, n_shards)).to(dev)
        assert torch.all(part[~mask] == 0), "a shard must not touch pixels it does not own"
        total += part
    assert segs == seg_f
    rgba = torch.zeros((h, w, 4), dtype=torch.uint8, device=dev)
    mean = torch.zeros((h, w, 3), dtype=torch.float32, device=dev)
    torch.cuda.synchronize()
    r.resolve_device(total.data_ptr(), scene.settings.samples_per_pixel, w, h, rgba.data_ptr(), mean.data_ptr(), 0)
    r.stats()
    torch.cuda.synchronize()
    assert np.array_equal(mean.cpu().numpy().view(np.uint32), mean_f.view(np.uint32))
    assert np.array_equal(rgba.cpu().numpy(), rgba_f)
    o_rgba, o_mean, _ = pyoracle.OracleScene(scene).render(w, h, seed, threads=0)
    assert np.array_equal(o_mean.view(np.uint32), mean_f.view(np.uint32)) and np.array_equal(o_rgba, rgba_f)


def test_render_entry_point_fills_render_target(gpu_renderer):
    """rt.render(target, scene): the reference's signature (render.rs:61)."""
    scene = benchmark_scene(4)
    target = rt.RenderTarget(64, 40)
    rt.render(target, scene, seed=5)
    assert target.request_redraw
    o_rgba, _, _ = pyoracle.OracleScene(scene).render(64, 40, 5, threads=0)
    assert np.array_equal(target.image(), o_rgba)


def test_full_size_frame_properties(gpu_renderer):
    """BASELINE's full frame size (1920x1080) at low spp: size-independent properties -- determinism, seed
    sensitivity, alpha = 255, sky pixels exactly the background, device-counted segments within the depth bound."""
    scene = benchmark_scene(2)
    r = gpu_renderer
    r.upload_scene(scene)
    a, _ = r.render(1920, 1080, seed=1)
    seg = r.stats().segments
    b, _ = r.render(1920, 1080, seed=1)
    c, _ = r.render(1920, 1080, seed=2)
    assert np.array_equal(a, b) and not np.array_equal(a, c)
    assert np.all(a[..., 3] == 255)
    assert np.all(a[:40, 600:1300, :3] == 197)  # sky: trunc(sqrt(0.6)*255)
    n = 1920 * 1080 * 2
    assert n <= seg <= n * scene.settings.max_ray_depth


def test_frame_pipeline_over_rccl_single_rank(gpu_renderer):
    """raytracingrust_amd.distributed.FramePipeline (what bench.py runs for N > 1) with a one-rank RCCL group:
    double-buffered zero -> integrate -> reduce -> resolve must reproduce the fused frame for every frame."""
    import os

    import torch
    import torch.distributed as dist

    from raytracingrust_amd.distributed import FramePipeline

    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29547")
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    created = not dist.is_initialized()
    if created:
        dist.init_process_group(backend="nccl", rank=0, world_size=1, device_id=dev)
    try:
        w, h = 120, 72
        scene = benchmark_scene(6)
        r = gpu_renderer
        r.upload_scene(scene)
        pipe = FramePipeline(r, w, h, scene.settings.samples_per_pixel, dev)
        frames = []
        for i in range(5):
            pipe.submit(100 + i)
            if i >= 3:  # read back the last frames one at a time (rgba8 is overwritten by the next resolve)
                pipe.wait()
                frames.append((100 + i, pipe.rgba8.cpu().numpy().copy()))
        pipe.wait()
        for seed, img in frames:
            ref, _ = r.render(w, h, seed=seed)
            assert np.array_equal(img, ref)
    finally:
        if created:
            dist.destroy_process_group()


def _fog_scene(bvh, spp=4, two_volumes=False):
    """src/gui.rs:98-103 (commented-out there): a big Isotropic fog volume around the scene, plus solids."""
    s = rt.Scene()
    s.settings.samples_per_pixel = spp
    s.settings.enable_bvh_tree = bvh
    s.settings.max_ray_depth = 8
    s.background = rt.GradientBackground([1.0, 1.0, 1.0], [0.5, 0.7, 1.0])
    ground = s.add_material(rt.Lambertian([0.5, 0.5, 0.5]))
    red = s.add_material(rt.Lambertian([0.8, 0.2, 0.2]))
    glass = s.add_material(rt.Dielectric(1.5))
    fog = s.add_material(rt.Isotropic([1.0, 1.0, 1.0]))
    smoke = s.add_material(rt.Isotropic([0.1, 0.1, 0.1]))
    s.add_object(rt.Sphere([0.0, -100.5, -1.0], 100.0, ground))
    s.add_object(rt.Sphere([0.6, 0.0, -1.2], 0.5, red))
    s.add_object(rt.Volume(rt.Sphere([0.0, 0.0, 0.0], 30.0, fog), 0.08))
    s.add_object(rt.Sphere([-0.7, 0.0, -1.0], 0.5, glass))
    if two_volumes:
        s.add_object(rt.Volume(rt.Sphere([-0.7, 0.9, -1.0], 0.4, smoke), 4.0))
        s.add_object(rt.Volume(rt.Sphere([0.2, 0.3, -0.6], 0.25, smoke), 9.0))
    return s


@pytest.mark.parametrize("bvh", [True, False])
@pytest.mark.parametrize("two", [False, True])
def test_volume_fog(gpu_renderer, bvh, two):
    """Volume::hit (volume.rs:34-73) draws a random number INSIDE world.hit, once per visit: parity needs the GPU
    traversal to visit Volume leaves exactly when and in the order the reference does (never culled, left first)."""
    c, st, o_st = run_both(gpu_renderer, _fog_scene(bvh, two_volumes=two), 96, 64, 31)
    assert st.segments == o_st.segments
    assert c["mismatched_pixels"] == 0, c
    assert c["max_u8_delta"] == 0


def _mesh_fog_scene(bvh, spp=4, with_sphere_fog=True):
    """Volume over a MESH boundary (volume.rs:20 `boundary: Box<dyn Object>`): smoke inside cube.obj and inside the monkey,
    next to solids, a mesh object and (optionally) a sphere-bounded fog -- the two boundary.hit calls of Volume::hit then
    walk the boundary mesh's own tree with the ranges [-inf, inf] and [t1 + tmin, inf] (mesh.rs:124-127,146-197)."""
    from raytracingrust_amd.configs import mesh_path

    s = rt.Scene()
    s.camera = rt.Camera([0.3, 0.6, 3.2], [0.0, 0.0, 0.0], [0.0, 1.0, 0.0], 55.0, 1.5)
    s.settings.samples_per_pixel = spp
    s.settings.enable_bvh_tree = bvh
    s.settings.max_ray_depth = 8
    s.background = rt.GradientBackground([1.0, 1.0, 1.0], [0.5, 0.7, 1.0])
    ground = s.add_material(rt.Lambertian([0.5, 0.5, 0.5]))
    red = s.add_material(rt.Lambertian([0.8, 0.2, 0.2]))
    smoke = s.add_material(rt.Isotropic([0.15, 0.15, 0.2]))
    mist = s.add_material(rt.Isotropic([0.9, 0.9, 0.9]))
    s.add_object(rt.Sphere([0.0, -101.0, 0.0], 100.0, ground))
    cube = rt.Mesh.from_file(mesh_path("cube.obj"), smoke)
    cube.positions = (cube.positions * np.float32(0.6) + np.array([-0.9, -0.2, 0.0], dtype=np.float32)).astype(np.float32)
    s.add_object(rt.Volume(cube, 3.0))
    s.add_object(rt.Sphere([1.3, -0.5, -0.3], 0.5, red))
    monkey = rt.Mesh.from_file(mesh_path("monkey.obj"), mist)
    monkey.positions = (monkey.positions * np.float32(0.8) + np.array([0.5, 0.1, 0.2], dtype=np.float32)).astype(np.float32)
    s.add_object(rt.Volume(monkey, 6.0))
    plane = rt.Mesh.from_file(mesh_path("plane.obj"), red)
    plane.positions = (plane.positions * np.float32(0.5) + np.array([0.0, 1.4, -0.5], dtype=np.float32)).astype(np.float32)
    s.add_object(plane)
    if with_sphere_fog:
        s.add_object(rt.Volume(rt.Sphere([0.0, 0.0, 0.0], 25.0, mist), 0.05))
    return s


@pytest.mark.parametrize("bvh", [True, False])
@pytest.mark.parametrize("sphere_fog", [False, True])
def test_volume_over_a_mesh_boundary(gpu_renderer, bvh, sphere_fog):
    """Bit-exact against the oracle, whose Volume takes any boundary through the reference's own Mesh::hit restatement."""
    c, st, o_st = run_both(gpu_renderer, _mesh_fog_scene(bvh, with_sphere_fog=sphere_fog), 96, 64, 17)
    assert st.segments == o_st.segments
    assert c["mismatched_pixels"] == 0, c
    assert c["max_u8_delta"] == 0


@pytest.mark.parametrize("tree", ["linear", "bvh", "simple"])
@pytest.mark.parametrize("w,h", [(1, 1), (1, 9), (9, 1), (2, 2), (8, 8), (9, 9)])
def test_degenerate_target_sizes(gpu_renderer, w, h, tree):
    """render.rs:88-89 divides by (W-1) and (H-1): a 1-pixel-wide target makes u or v inf/NaN.  The reference does
    not guard it; neither do we -- but the GPU must go wrong exactly like the oracle (NaN where it has NaN).
    In the tree worlds every sphere then reports a NaN distance and the nested merges (bvh.rs:181-198, NaN -> right) pick
    the last hit leaf: an Emission sphere there returns a finite colour, so the choice is visible."""
    scene = simple_scene(3) if tree == "simple" else benchmark_scene(3, enable_bvh_tree=(tree == "bvh"))
    r = gpu_renderer
    r.upload_scene(scene)
    rgba, mean = r.render(w, h, seed=13, want_rgba8=True, want_mean=True)
    st = r.stats()
    o_rgba, o_mean, o_st = pyoracle.OracleScene(scene).render(w, h, 13, threads=0)
    assert st.segments == o_st.segments
    both_nan = np.isnan(mean) & np.isnan(o_mean)
    same = (mean.view(np.uint32) == o_mean.view(np.uint32)) | both_nan
    assert np.all(same), (mean, o_mean)
    assert np.array_equal(rgba, o_rgba)


def test_empty_scene_and_extreme_settings(gpu_renderer):
    r = gpu_renderer
    s = rt.Scene()  # no objects: BvhNode::None
    s.settings.samples_per_pixel = 3
    for bvh in (True, False):
        s.settings.enable_bvh_tree = bvh
        c, st, o_st = run_both(r, s, 24, 16, 1)
        assert c["mismatched_pixels"] == 0 and st.segments == o_st.segments == 24 * 16 * 3
    b = benchmark_scene(1, max_ray_depth=255)  # u8::MAX
    c, st, o_st = run_both(r, b, 32, 24, 2)
    assert c["mismatched_pixels"] == 0 and st.segments == o_st.segments
    b = benchmark_scene(0)  # zero samples: 0/0 = NaN -> `as u8` = 0 (color.rs:34-41)
    r.upload_scene(b)
    rgba, mean = r.render(16, 8, seed=1, want_rgba8=True, want_mean=True)
    assert np.all(np.isnan(mean)) and np.all(rgba[..., :3] == 0) and np.all(rgba[..., 3] == 255)
    with pytest.raises(rt.RtrError):
        r.render(0, 8)
    bad = benchmark_scene(1, clamp_indirect=float("nan"))
    with pytest.raises(rt.RtrError):
        r.render(8, 8, scene=bad)
    bad = benchmark_scene(1, max_ray_depth=256)
    with pytest.raises(rt.RtrError):
        r.render(8, 8, scene=bad)


@pytest.mark.parametrize("n_shards", [2, 3, 8])
def test_sample_sharded_frame_matches_within_float_sum_order(gpu_renderer, n_shards):
    """The multi-GPU decomposition bench.py uses (rtr_render_samples_device): every shard integrates all pixels for
    its contiguous share of the samples.  Draws are identical to the single launch (streams are keyed by the absolute
    sample index: the segment counts add up exactly); only the float addition order of the per-pixel sums differs,
    so the frame matches to ~1e-7 relative and the RGBA8 image to +-1."""
    import torch

    from raytracingrust_amd.distributed import sample_range

    w, h, seed = 96, 56, 77
    scene = simple_scene(13)  # 13 spp: uneven shares
    r = gpu_renderer
    r.upload_scene(scene)
    rgba_f, mean_f = r.render(w, h, seed=seed, want_rgba8=True, want_mean=True)
    seg_f = r.stats().segments
    dev = torch.device("cuda", 0)
    total = torch.zeros((h, w, 3), dtype=torch.float32, device=dev)
    segs, covered = 0, 0
    for shard in range(n_shards):
        begin, count = sample_range(shard, n_shards, 13)
        assert begin == covered
        covered += count
        part = torch.full((h, w, 3), 5.0, dtype=torch.float32, device=dev)  # every pixel must be overwritten
        torch.cuda.synchronize()
        r.render_samples_device(w, h, seed, begin, count, part.data_ptr(), 0)
        segs += r.stats().segments
        total += part
    assert covered == 13 and segs == seg_f
    rgba = torch.zeros((h, w, 4), dtype=torch.uint8, device=dev)
    mean = torch.zeros((h, w, 3), dtype=torch.float32, device=dev)
    torch.cuda.synchronize()
    r.resolve_device(total.data_ptr(), 13, w, h, rgba.data_ptr(), mean.data_ptr(), 0)
    r.stats()
    c = compare(mean.cpu().numpy(), rgba.cpu().numpy(), mean_f, rgba_f)
    assert max(c["rel_l2"]) <= 1e-6 and c["max_u8_delta"] <= 1, c
    with pytest.raises(rt.RtrError):
        r.render_samples_device(w, h, seed, 10, 4, total.data_ptr(), 0)  # 10 + 4 > 13


@pytest.mark.parametrize("config,spp", [("c1", 2), ("c4", 1), ("c3l", 1), ("c3", 1), ("c5", 1)])
def test_full_size_frames_against_the_oracle(gpu_renderer, config, spp):
    """BASELINE's frame sizes for the tree worlds -- 1920x1080: C1 (simple.json), C4 (house + plane, mix material, HDRI), C3
    (glass monkey + plane, HDRI: the benched configuration) and its Lambertian variant; 3840x2160: C5 (simple.json + HDRI:
    129 600 tiles, 8.3 M (sample, pixel) slots per sample) -- at low spp, whole frame against the oracle (polynomial trig
    mode)."""
    from raytracingrust_amd.configs import baseline_config

    scene, w, h = baseline_config(config, spp)
    c, st, o_st = run_both(gpu_renderer, scene, w, h, 4242)
    assert st.segments == o_st.segments
    assert max(c["rel_l2"]) <= TOL and c["max_u8_delta"] <= 1, c
    assert c["mismatched_pixels"] == 0, c  # culled (kCullMargin) vs unculled walk: not one pixel at 1080p


def _random_scene(rng, with_mesh, with_volume):
    s = rt.Scene()
    s.settings.samples_per_pixel = int(rng.integers(1, 6))
    s.settings.max_ray_depth = int(rng.integers(1, 9))
    s.settings.clamp_indirect = float(rng.choice([0.5, 2.0, 10.0]))
    s.settings.enable_bvh_tree = bool(rng.integers(0, 2))
    s.camera = rt.Camera(rng.uniform(-1, 1, 3) + [0, 0.5, 3], rng.uniform(-0.5, 0.5, 3), [0, 1, 0], float(rng.uniform(30, 100)),
                         float(rng.uniform(0.8, 2.0)))
    bgk = int(rng.integers(0, 3))
    if bgk == 0:
        s.background = rt.UniformBackground(rng.uniform(0, 1.5, 3))
    elif bgk == 1:
        s.background = rt.GradientBackground(rng.uniform(0, 1, 3), rng.uniform(0, 1, 3))
    else:
        s.background = rt.SkyMap(rng.uniform(0, 3, (int(rng.integers(1, 9)), int(rng.integers(1, 17)), 3)).astype(np.float32))

    def material(depth=0):
        k = int(rng.integers(0, 6 if depth < 2 else 5))
        if k == 0:
            return rt.Lambertian(rng.uniform(0, 1, 3))
        if k == 1:
            return rt.Metal(rng.uniform(0, 1, 3), float(rng.choice([0.0, 0.05, 0.5, 1.0])))
        if k == 2:
            return rt.Emission(rng.uniform(0, 1, 3), float(rng.uniform(0.5, 20)))
        if k == 3:
            return rt.Dielectric(float(rng.uniform(1.1, 2.4)))
        if k == 4:
            return rt.Isotropic(rng.uniform(0, 1, 3))
        return rt.MixMaterial(material(depth + 1), material(depth + 1), float(rng.uniform(0, 1)))

    mats = [s.add_material(material()) for _ in range(int(rng.integers(1, 7)))]
    pick = lambda: mats[int(rng.integers(0, len(mats)))]  # noqa: E731
    for _ in range(int(rng.integers(0, 9))):
        s.add_object(rt.Sphere(rng.uniform(-2, 2, 3), float(rng.uniform(0.05, 1.2)), pick()))
    if with_mesh:
        for _ in range(int(rng.integers(1, 3))):
            nt = int(rng.integers(1, 40))
            pos = (rng.uniform(-1.5, 1.5, (nt, 1, 3)) + rng.uniform(-0.5, 0.5, (nt, 3, 3))).reshape(-1, 3).astype(np.float32)
            nrm = rng.normal(size=(nt * 3, 3)).astype(np.float32)
            nrm /= np.linalg.norm(nrm, axis=1, keepdims=True)
            s.add_object(rt.Mesh(pos, nrm, np.arange(nt * 3, dtype=np.uint32), pick()))
    if with_volume:
        for _ in range(int(rng.integers(1, 3))):
            s.add_object(rt.Volume(rt.Sphere(rng.uniform(-1, 1, 3), float(rng.uniform(0.3, 3.0)), pick()), float(rng.uniform(0.05, 5))))
    if rng.integers(0, 4) == 0:
        s.settings.mode = rt.RenderMode(int(rng.integers(1, 4)))
    return s


@pytest.mark.parametrize("case", range(36))
def test_random_scenes(gpu_renderer, case):
    """Seeded random scenes -- every material kind incl. nested mixes, 0..8 spheres, random triangle soups, volumes, all
    three backgrounds, both worlds, all render modes, odd frame sizes -- against the oracle.  Sphere / volume scenes must
    be bit-identical; scenes with meshes may differ in a handful of pixels (culled vs unculled walk)."""
    rng = np.random.default_rng(1000 + case)
    with_mesh, with_volume = case % 3 == 1, case % 3 == 2
    scene = _random_scene(rng, with_mesh, with_volume)
    w, h = int(rng.integers(5, 70)), int(rng.integers(5, 50))
    c, st, o_st = run_both(gpu_renderer, scene, w, h, int(rng.integers(0, 2**62)))
    assert st.segments == o_st.segments, (case, st.segments, o_st.segments)
    assert c["max_u8_delta"] <= 1
    finite = max(c["rel_l2"]) == max(c["rel_l2"])  # NaN radiance (e.g. 0 * inf emission chains) compares as NaN both sides
    if finite:
        assert max(c["rel_l2"]) <= TOL, (case, c)
    assert c["mismatched_pixels"] == 0, (case, c)


@pytest.mark.parametrize("builder,w,h", [(lambda: house_scene(2), 192, 108), (lambda: monkey_scene(3), 160, 90),
                                         (lambda: monkey_scene(2, material=rt.Dielectric(1.5), bvh=False), 96, 54)])
def test_sah_mesh_trees_render_the_same_frame(builder, w, h):
    """Option mesh_tree = "sah" (SURVEY 8f N1): mesh subtrees rebuilt with a surface-area heuristic.  The walk visits
    about half the nodes but must find the same closest hits: same segment count as the oracle (no path changes
    length), frame within tolerance, at most a handful of last-bit pixel differences (box-boundary cases)."""
    scene = builder()
    with rt.Renderer(0) as r:
        r.set_option("mesh_tree", "sah")
        r.upload_scene(scene)
        rgba, mean = r.render(w, h, seed=12, want_rgba8=True, want_mean=True)
        st = r.stats()
        sah_height = r.scene_info().tree_height
        r.set_option("mesh_tree", "reference")
        r.upload_scene(scene)
        assert r.scene_info().tree_height <= sah_height <= 2 * r.scene_info().tree_height  # depth-limited
        with pytest.raises(rt.RtrError):
            r.set_option("mesh_tree", "octree")
    o_rgba, o_mean, o_st = pyoracle.OracleScene(scene).render(w, h, 12, threads=0)
    c = compare(mean, rgba, o_mean, o_rgba)
    assert st.segments == o_st.segments
    assert max(c["rel_l2"]) <= TOL and c["max_u8_delta"] <= 1, c
    assert c["mismatched_pixels"] <= 5, c


@pytest.mark.parametrize("builder,w,h", [(lambda: house_scene(5), 157, 83), (lambda: simple_scene(7), 61, 130)])
def test_claimed_work_blocks_do_not_change_a_bit(builder, w, h):
    """The samples kernel's waves claim blocks of (tile, block_samples consecutive samples) from a device counter.  Which
    wave and lane renders a (sample, pixel) pair must not matter: block sizes that divide the sample count, that do not,
    and that exceed it -- on frames with ragged tile edges -- give the oracle's frame bit for bit."""
    scene = builder()
    o_rgba, o_mean, o_st = pyoracle.OracleScene(scene).render(w, h, 31, threads=0)
    with rt.Renderer(0) as r:
        r.set_option("tree_kernel", "samples")
        r.upload_scene(scene)
        for bs in ("1", "2", "3", "8", "1024"):
            r.set_option("block_samples", bs)
            rgba, mean = r.render(w, h, seed=31, want_rgba8=True, want_mean=True)
            assert r.stats().segments == o_st.segments, bs
            assert np.array_equal(rgba, o_rgba), bs
            assert np.array_equal(mean.view(np.uint32), o_mean.view(np.uint32)), bs


@pytest.mark.parametrize("builder,w,h", [(lambda: house_scene(2), 160, 90), (lambda: simple_scene(4), 128, 128),
                                         (lambda: monkey_scene(3), 96, 54)])
def test_stack_entry_forms_render_the_same_frame(builder, w, h):
    """The samples kernel keeps a traversal-stack entry in one word (4 flag bits, 16-bit index, 12 bits of entry
    distance) when the tree has at most 65 536 nodes and leaves, else in a 32-bit ref plus a 16-bit distance (option
    stack_entries = "wide" forces that form).  The distance is only a lower bound used to skip boxes the best hit has
    overtaken, so both forms -- and the oracle's unculled walk -- must give the same frame; LDS per wave must shrink."""
    scene = builder()
    with rt.Renderer(0) as r:
        r.set_option("tree_kernel", "samples")
        r.upload_scene(scene)
        rgba_p, mean_p = r.render(w, h, seed=77, want_rgba8=True, want_mean=True)
        st_p = r.stats()
        r.set_option("stack_entries", "wide")
        rgba_w, mean_w = r.render(w, h, seed=77, want_rgba8=True, want_mean=True)
        st_w = r.stats()
        with pytest.raises(rt.RtrError):
            r.set_option("stack_entries", "narrow")
    assert st_p.segments == st_w.segments
    assert st_p.lds_bytes < st_w.lds_bytes
    assert np.array_equal(rgba_p, rgba_w)
    assert np.array_equal(mean_p.view(np.uint32), mean_w.view(np.uint32))
    o_rgba, o_mean, o_st = pyoracle.OracleScene(scene).render(w, h, 77, threads=0)
    assert o_st.segments == st_p.segments
    assert np.array_equal(rgba_p, o_rgba)


@pytest.mark.parametrize("builder,w,h", [(lambda: house_scene(2), 160, 90), (lambda: simple_scene(4), 128, 128),
                                         (lambda: monkey_scene(2, bvh=False), 96, 54),
                                         (lambda: benchmark_scene(4, enable_bvh_tree=True), 128, 72)])
def test_leaf_pairs_do_not_change_a_bit(builder, w, h):
    """Option leaf_pairs (default "on") folds Split(leaf L, leaf L+1) nodes into their parents: the same box and leaf
    tests in the same order, so the frame must be bit-identical to the unfolded tree's and to the oracle's."""
    scene = builder()
    with rt.Renderer(0) as r:
        r.upload_scene(scene)
        nodes_on = r.scene_info().n_nodes
        rgba_on, mean_on = r.render(w, h, seed=21, want_rgba8=True, want_mean=True)
        seg_on = r.stats().segments
        r.set_option("leaf_pairs", "off")
        r.upload_scene(scene)
        assert r.scene_info().n_nodes == nodes_on  # scene_info reports the reference's Split count either way
        rgba_off, mean_off = r.render(w, h, seed=21, want_rgba8=True, want_mean=True)
        assert r.stats().segments == seg_on
        with pytest.raises(rt.RtrError):
            r.set_option("leaf_pairs", "maybe")
    assert np.array_equal(rgba_on, rgba_off)
    assert np.array_equal(mean_on.view(np.uint32), mean_off.view(np.uint32))
    o_rgba, o_mean, o_st = pyoracle.OracleScene(scene).render(w, h, 21, threads=0)
    assert o_st.segments == seg_on
    c = compare(mean_on, rgba_on, o_mean, o_rgba)
    assert max(c["rel_l2"]) <= TOL and c["max_u8_delta"] <= 1, c


def test_sah_scene_tree_renders_the_same_frame():
    """Option scene_tree = "sah": the scene-level tree over simple.json's 102 spheres rebuilt with SAH.  Sphere boxes
    are exact supersets in either tree, so the frame stays bit-identical apart from box-boundary grazes."""
    scene = simple_scene(6)
    w, h = 160, 160
    with rt.Renderer(0) as r:
        r.set_option("scene_tree", "sah")
        r.upload_scene(scene)
        rgba, mean = r.render(w, h, seed=3, want_rgba8=True, want_mean=True)
        st = r.stats()
    o_rgba, o_mean, o_st = pyoracle.OracleScene(scene).render(w, h, 3, threads=0)
    c = compare(mean, rgba, o_mean, o_rgba)
    assert st.segments == o_st.segments
    assert max(c["rel_l2"]) <= TOL and c["max_u8_delta"] <= 1 and c["mismatched_pixels"] <= 3, c
    # a scene with a Volume keeps the reference scene tree (its random draws depend on the visit order)
    fog = _fog_scene(True, two_volumes=True)
    with rt.Renderer(0) as r:
        r.set_option("scene_tree", "sah")
        c, st, o_st = run_both(r, fog, 64, 48, 9)
    assert c["mismatched_pixels"] == 0 and st.segments == o_st.segments


def test_concurrent_calls_on_one_context_serialise():
    """include/rtr_hip.h: calls on one context take its internal mutex.  Four threads render different seeds through
    the same Renderer at once (ctypes drops the GIL inside the call); every frame must equal its single-threaded
    twin."""
    import threading

    scene = benchmark_scene(2)
    w, h = 96, 64
    with rt.Renderer(0) as r:
        r.upload_scene(scene)
        want = {seed: r.render(w, h, seed=seed, want_rgba8=True, want_mean=True) for seed in range(4)}
        got, errors = {}, []

        def work(seed):
            try:
                for _ in range(3):
                    got[seed] = r.render(w, h, seed=seed, want_rgba8=True, want_mean=True)
            except Exception as e:  # noqa: BLE001
                errors.append(e)

        threads = [threading.Thread(target=work, args=(seed,)) for seed in range(4)]
        for t in threads:
            t.start()
        for t in threads:
            t.join()
    assert not errors, errors
    for seed in range(4):
        assert np.array_equal(got[seed][0], want[seed][0])
        assert np.array_equal(got[seed][1].view(np.uint32), want[seed][1].view(np.uint32))


def _tiny_tree_scene(n):
    """BVH world over 0..3 spheres: BvhNode::None, a bare leaf as root (no box at all, bvh.rs:54), a leaf pair, and
    a Split over a leaf and a pair."""
    s = rt.Scene()
    s.settings.samples_per_pixel = 5
    s.settings.enable_bvh_tree = True
    s.background = rt.GradientBackground([1.0, 1.0, 1.0], [0.5, 0.7, 1.0])
    m = s.add_material(rt.Lambertian([0.7, 0.3, 0.3]))
    g = s.add_material(rt.Metal([0.8, 0.8, 0.8], 0.1))
    for c, r, mat in [([0.0, 0.0, -1.0], 0.5, m), ([0.0, -100.5, -1.0], 100.0, g), ([1.0, 0.0, -1.2], 0.4, g)][:n]:
        s.add_object(rt.Sphere(c, r, mat))
    return s


@pytest.mark.parametrize("builder,w,h", [(lambda: house_scene(2), 160, 90), (lambda: simple_scene(4), 128, 128),
                                         (lambda: monkey_scene(3, material=rt.Dielectric(1.5)), 128, 72),
                                         (lambda: benchmark_scene(4, enable_bvh_tree=True), 128, 72),
                                         (lambda: _fog_scene(True, two_volumes=True), 96, 64),
                                         (lambda: _mesh_fog_scene(True, spp=3), 72, 48),
                                         (lambda: _tiny_tree_scene(0), 40, 24), (lambda: _tiny_tree_scene(1), 40, 24),
                                         (lambda: _tiny_tree_scene(2), 40, 24), (lambda: _tiny_tree_scene(3), 40, 24)])
def test_tree_kernels_agree_bit_for_bit(builder, w, h):
    """Option tree_kernel: the wave-synchronous walk (k_render), the resumable per-lane walk (k_render_bvh), the
    sample-parallel walk (k_render_bvh_sp) and the streaming wavefront (k_wf_*: paths in device memory, one kernel per
    stage and depth) execute the same operations per path in the same order, so their frames must be identical to the
    last bit -- and equal the oracle's."""
    scene = builder()
    frames = {}
    with rt.Renderer(0) as r:
        r.upload_scene(scene)
        for kernel in ("wave", "resumable", "auto", "samples", "stream"):
            r.set_option("tree_kernel", kernel)
            rgba, mean = r.render(w, h, seed=77, want_rgba8=True, want_mean=True)
            frames[kernel] = (rgba, mean.view(np.uint32).copy(), r.stats().segments)
        with pytest.raises(rt.RtrError):
            r.set_option("tree_kernel", "fast")
    for kernel in ("resumable", "auto", "samples", "stream"):
        assert frames[kernel][2] == frames["wave"][2], kernel
        assert np.array_equal(frames[kernel][0], frames["wave"][0]) and np.array_equal(frames[kernel][1], frames["wave"][1])
    o_rgba, o_mean, o_st = pyoracle.OracleScene(scene, trig_mode=pyoracle.TRIG_POLY).render(w, h, 77, threads=0)
    assert o_st.segments == frames["wave"][2]
    c = compare(frames["resumable"][1].view(np.float32), frames["resumable"][0], o_mean, o_rgba)
    assert max(c["rel_l2"]) <= TOL and c["max_u8_delta"] <= 1, c


@pytest.mark.parametrize("builder,w,h", [(lambda: benchmark_scene(13), 97, 61), (lambda: house_scene(5), 96, 54),
                                         (lambda: simple_scene(7), 64, 64), (lambda: monkey_scene(6, bvh=False), 64, 36),
                                         (lambda: _fog_scene(True, spp=9, two_volumes=True), 48, 32)])
def test_sample_chunks_do_not_change_a_bit(builder, w, h):
    """Option sample_chunks: one launch renders every pixel's samples in consecutive pieces (chunk-major workgroups,
    running sums handed over through memory).  Same additions in the same order: the frame and the segment count are
    identical for every chunk count, including counts that do not divide spp and more chunks than samples."""
    scene = builder()
    frames = []
    with rt.Renderer(0) as r:
        r.upload_scene(scene)
        for chunks in ("1", "auto", "2", "3", "4", "7", "64"):
            r.set_option("sample_chunks", chunks)
            rgba, mean = r.render(w, h, seed=19, want_rgba8=True, want_mean=True)
            frames.append((rgba, mean.view(np.uint32).copy(), r.stats().segments))
        # the pieces' lengths: the linear taper of an explicit count, or given proportions (what "auto" picks on large frames)
        for name, value in (("debug_chunk_taper", "0"), ("debug_chunk_taper", "7"), ("debug_chunk_lengths", "24:7:1"),
                            ("debug_chunk_lengths", "7:1"), ("debug_chunk_lengths", "1:1:1:1:1:1:1:40"), ("debug_chunk_lengths", "1:900")):
            r.set_option(name, value)
            rgba, mean = r.render(w, h, seed=19, want_rgba8=True, want_mean=True)
            frames.append((rgba, mean.view(np.uint32).copy(), r.stats().segments))
        for bad in ("0", "4097", "many", ""):
            with pytest.raises(rt.RtrError):
                r.set_option("sample_chunks", bad)
        for bad in ("", "0:1", "1:2:3:4:5:6:7:8:9", "3:x"):
            with pytest.raises(rt.RtrError):
                r.set_option("debug_chunk_lengths", bad)
    for rgba, mean_bits, seg in frames[1:]:
        assert seg == frames[0][2]
        assert np.array_equal(rgba, frames[0][0]) and np.array_equal(mean_bits, frames[0][1])
    o_rgba, o_mean, o_st = pyoracle.OracleScene(scene, trig_mode=pyoracle.TRIG_POLY).render(w, h, 19, threads=0)
    assert o_st.segments == frames[0][2]


def test_sample_chunks_in_the_sharded_paths(gpu_renderer):
    """Chunks inside the sample-sharded and tile-sharded launches (what each rank of the multi-GPU path runs): the
    float4 sums must equal the unchunked ones bit for bit."""
    import torch

    r = gpu_renderer
    scene = benchmark_scene(24)
    w, h = 120, 72
    r.upload_scene(scene)
    dev = torch.device("cuda", 0)
    out = {}
    try:
        for chunks in ("1", "3"):
            r.set_option("sample_chunks", chunks)
            a = torch.zeros((h, w, 3), dtype=torch.float32, device=dev)
            b = torch.zeros((h, w, 3), dtype=torch.float32, device=dev)
            r.render_samples_device(w, h, 5, 6, 11, a.data_ptr(), scene=scene)   # samples [6, 17)
            r.render_tiles_device(w, h, 5, 1, 3, b.data_ptr(), scene=scene)      # shard 1 of 3
            torch.cuda.synchronize()
            out[chunks] = (a.cpu().numpy().view(np.uint32).copy(), b.cpu().numpy().view(np.uint32).copy())
    finally:
        r.set_option("sample_chunks", "auto")
    assert np.array_equal(out["1"][0], out["3"][0]) and np.array_equal(out["1"][1], out["3"][1])


def test_automatic_chunks_on_a_full_size_frame(gpu_renderer):
    """What "sample_chunks" = "auto" picks only on large frames (sphere world: pieces in the proportions 24:7:1 from 96 samples
    on, 7:1 from 28, with the hand-over made of device-scope atomic accesses): a rank's share of the headline frame -- 1920x1080,
    a sample range that does not start at 0 -- must give the sums of the one-piece launch bit for bit, and so must the frame."""
    import torch

    r = gpu_renderer
    w, h = 1920, 1080
    dev = torch.device("cuda", 0)
    try:
        for spp, first, count in ((128, 32, 96), (64, 17, 40)):
            scene = benchmark_scene(spp)
            r.upload_scene(scene)
            sums, frames = [], []
            for chunks in ("1", "auto", "5"):
                r.set_option("sample_chunks", chunks)
                a = torch.zeros((h, w, 3), dtype=torch.float32, device=dev)
                r.render_samples_device(w, h, 77, first, count, a.data_ptr(), scene=scene)
                torch.cuda.synchronize()
                sums.append(a.cpu().numpy().view(np.uint32).copy())
                rgba = r.render(w, h, seed=77, want_rgba8=True, want_mean=False)[0]
                frames.append((rgba, r.stats().segments))
            for k in (1, 2):
                assert np.array_equal(sums[0], sums[k])
                assert frames[0][1] == frames[k][1] and np.array_equal(frames[0][0], frames[k][0])
    finally:
        r.set_option("sample_chunks", "auto")


def test_sample_parallel_kernel_in_batches_and_sample_shards(gpu_renderer):
    """tree_kernel = "samples" / "stream": per-sample radiance goes through device memory and is added per pixel in index order.
    The sample-sharded launches (what a rank of the multi-GPU path runs) must give the same float4 sums as the resumable
    kernel, bit for bit, also when the sample range does not start at 0 and with several chunks."""
    import torch

    r = gpu_renderer
    scene = house_scene(12)
    w, h = 104, 60
    r.upload_scene(scene)
    dev = torch.device("cuda", 0)
    out = {}
    try:
        for kernel, chunks, batch in (("resumable", "1", "0"), ("samples", "auto", "0"), ("samples", "5", "0"), ("samples", "2", "3"),
                                      ("samples", "auto", "1"), ("stream", "auto", "0"), ("stream", "auto", "3"), ("stream", "auto", "1")):
            r.set_option("tree_kernel", kernel)
            r.set_option("sample_chunks", chunks)
            r.set_option("samples_batch", batch)
            a = torch.zeros((h, w, 3), dtype=torch.float32, device=dev)
            r.render_samples_device(w, h, 9, 3, 8, a.data_ptr(), scene=scene)  # samples [3, 11)
            torch.cuda.synchronize()
            rgba, mean = r.render(w, h, seed=9, want_rgba8=True, want_mean=True)  # the fused path, same options
            out[(kernel, chunks, batch)] = (a.cpu().numpy().view(np.uint32).copy(), r.stats().segments, rgba, mean.view(np.uint32).copy())
    finally:
        r.set_option("tree_kernel", "auto")
        r.set_option("sample_chunks", "auto")
        r.set_option("samples_batch", "0")
    base = out[("resumable", "1", "0")]
    for key, val in out.items():
        assert val[1] == base[1], key
        assert np.array_equal(val[0], base[0]) and np.array_equal(val[2], base[2]) and np.array_equal(val[3], base[3]), key


# ---- robustness of the asynchronous entry points ----------------------------------------------------------------
def _fresh_renderer(scene):
    r = rt.Renderer(0)
    r.upload_scene(scene)
    return r


def test_device_fault_word_fails_the_next_synchronising_call_once():
    """The device's sticky fault word (set by a sample chunk that gave up waiting; here by option debug_fault) must turn
    into RTR_ERR_HIP at every entry point that synchronises -- the device-resident render calls cannot report it -- and
    be cleared by that report."""
    import torch

    from raytracingrust_amd._capi import RtrError

    scene = benchmark_scene(4)
    w, h = 64, 48
    out = torch.zeros((h, w, 4), dtype=torch.uint8, device="cuda:0")
    for reader in ("stats", "kernel_times_ms", "segment_counts", "integrate_times_ms"):
        r = _fresh_renderer(scene)
        try:
            r.set_option("debug_fault", "on")
            r.render_frame_device(w, h, 1, out.data_ptr(), 0)  # asynchronous: returns RTR_OK
            r.set_option("debug_fault", "off")
            with pytest.raises(RtrError) as e:
                getattr(r, reader)()
            assert e.value.code == -3 and "fault" in str(e.value)
            getattr(r, reader)()  # reported once, then clean
            r.render_frame_device(w, h, 1, out.data_ptr(), 0)
            assert r.stats().segments > 0
        finally:
            r.close()
    r = _fresh_renderer(scene)
    try:
        r.set_option("debug_fault", "on")
        with pytest.raises(RtrError):
            r.render(w, h, seed=1)  # the host-output entry point synchronises itself
    finally:
        r.close()


def test_sample_chunk_that_gives_up_is_reported():
    """A frame of few tiles cut into chunks has successors resident next to their predecessors, so a chunk really
    waits; with a zero timeout it gives up at once and the frame must be reported invalid.  With the default bound the
    same launch is valid and bit-identical to the unchunked one."""
    from raytracingrust_amd._capi import RtrError

    scene = benchmark_scene(64)
    r = _fresh_renderer(scene)
    try:
        r.set_option("sample_chunks", "1")
        ref, _ = r.render(64, 64, seed=9)
        r.set_option("sample_chunks", "4")
        got, _ = r.render(64, 64, seed=9)
        assert np.array_equal(ref, got)
        r.set_option("chunk_timeout_ms", "0")
        with pytest.raises(RtrError) as e:
            for _ in range(8):  # (dispatch timing decides whether a particular launch has to wait at all)
                r.render(64, 64, seed=9)
        assert "timed out" in str(e.value)
        r.set_option("chunk_timeout_ms", "10000")
        got, _ = r.render(64, 64, seed=9)
        assert np.array_equal(ref, got)
    finally:
        r.close()


def test_frames_of_different_seeds_are_not_pixel_permutations(gpu_renderer):
    """Both halves of the seed are hashed into both stream keys: with `pixel ^ seed_lo` alone the frame of seed s ^ d
    was the frame of seed s with pixels p and p ^ d exchanged.  A scene of a uniform background only
    isolates the camera jitter... every pixel is then equal; use the benchmark scene at 1 spp instead: the multiset of
    pixel values must differ between seeds, and so must the frame under the XOR permutation."""
    scene = benchmark_scene(1)
    gpu_renderer.upload_scene(scene)
    w, h = 64, 64
    _, a = gpu_renderer.render(w, h, seed=0x1234, want_rgba8=False, want_mean=True)
    for d in (1, 2, 64, 0x1235 ^ 0x1234):
        _, b = gpu_renderer.render(w, h, seed=0x1234 ^ d, want_rgba8=False, want_mean=True)
        idx = (np.arange(w * h) ^ d) % (w * h)
        assert not np.array_equal(a.reshape(-1, 3)[idx], b.reshape(-1, 3))
        assert not np.array_equal(np.sort(a.reshape(-1, 3), axis=0), np.sort(b.reshape(-1, 3), axis=0))


# ---- N1: device-side BVH builder ------------------------------------------------------------------------------------
@pytest.mark.parametrize("builder,w,h", [(lambda: house_scene(2), 160, 90), (lambda: monkey_scene(3, material=rt.Dielectric(1.5)), 128, 72),
                                         (lambda: monkey_scene(2, bvh=False), 96, 54)])
def test_device_built_mesh_trees_render_the_same_frame(builder, w, h):
    """Option mesh_tree = "lbvh" (csrc/rtr_lbvh.hip): the mesh subtrees are linear BVHs built on the GPU (Morton sort,
    Karras radix tree, bottom-up boxes) over the reference's per-triangle boxes.  A different tree must find the same
    closest hits: same segment count as the oracle, frame within tolerance, at most a handful of last-bit pixel
    differences (box-boundary grazes, as with the SAH trees), for every tree kernel; and the tree it reports must be a
    valid one (n - 1 inner nodes per mesh, every leaf reachable once: checked by the frame being right)."""
    scene = builder()
    o_rgba, o_mean, o_st = pyoracle.OracleScene(scene, trig_mode=pyoracle.TRIG_POLY).render(w, h, 12, threads=0)
    with rt.Renderer(0) as r:
        r.upload_scene(scene)
        ref_info = r.scene_info()
        r.set_option("mesh_tree", "lbvh")
        r.upload_scene(scene)
        info = r.scene_info()
        assert info.n_leaves == ref_info.n_leaves and info.n_nodes == ref_info.n_nodes  # binary trees over the same leaves
        assert ref_info.tree_height <= info.tree_height <= 64
        frames = {}
        for kernel in ("auto", "wave", "resumable", "samples", "stream"):
            r.set_option("tree_kernel", kernel)
            rgba, mean = r.render(w, h, seed=12, want_rgba8=True, want_mean=True)
            st = r.stats()
            assert st.segments == o_st.segments, kernel
            c = compare(mean, rgba, o_mean, o_rgba)
            assert max(c["rel_l2"]) <= TOL and c["max_u8_delta"] <= 1 and c["mismatched_pixels"] <= 5, (kernel, c)
            frames[kernel] = mean.view(np.uint32).copy()
        for kernel in frames:
            assert np.array_equal(frames[kernel], frames["auto"]), kernel  # one tree, one frame, whichever kernel walks it
    with rt.Renderer(-1) as host_only:  # the device builder needs a device: a host-only context refuses
        host_only.set_option("mesh_tree", "lbvh")
        with pytest.raises(rt.RtrError):
            host_only.upload_scene(scene)


def test_sample_buffers_shrink_when_the_device_is_short_of_memory():
    """the per-sample radiance buffer (and the streaming kernels' path records) must not turn a shortage of
    device memory into an error: a batch whose allocation fails is halved and retried, and any batch size gives the same
    frame.  debug_alloc_limit makes allocations above a size fail."""
    scene = house_scene(8)
    w, h = 96, 54
    with rt.Renderer(0) as r:
        r.upload_scene(scene)
        for kernel in ("samples", "stream"):
            r.set_option("tree_kernel", kernel)
            ref_rgba, ref_mean = r.render(w, h, seed=4, want_rgba8=True, want_mean=True)
            seg = r.stats().segments
        with rt.Renderer(0) as small:
            small.upload_scene(scene)
            small.set_option("tree_kernel", kernel)
            # room for 3 samples of radiance ("samples") / 2 samples of path records ("stream"), not for all 8
            small.set_option("debug_alloc_limit", str(w * h * 16 * 3 if kernel == "samples" else 1_000_000))
            rgba, mean = small.render(w, h, seed=4, want_rgba8=True, want_mean=True)
            assert small.stats().segments == seg
            assert np.array_equal(rgba, ref_rgba) and np.array_equal(mean.view(np.uint32), ref_mean.view(np.uint32))
            small.set_option("debug_alloc_limit", "1000")  # not even one sample fits: a clean error, not a crash
            with pytest.raises(rt.RtrError):
                small.render(w + 8, h, seed=4)


# ---- N2: opt-in bilinear SkyMap filter ---------------------------------------------------------------------------------
@pytest.mark.parametrize("bvh", [True, False])
def test_bilinear_sky_filter_matches_the_oracles_restatement(bvh):
    """Option sky_filter = "bilinear"  is an extension -- SkyMap::sample takes the nearest texel -- with its
    arithmetic restated in the oracle: bit-exact for every tree kernel (the sample-parallel kernel's deferred sky lookup
    stands down), different from the nearest-texel frame, and back to it when the option is reset."""
    from raytracingrust_amd.hdri import synthetic_sky

    scene = monkey_scene(3, material=rt.Dielectric(1.5), background=rt.SkyMap(synthetic_sky(96, 48, sun_radius=0.1)), bvh=bvh)
    w, h = 96, 54
    o_rgba, o_mean, o_st = pyoracle.OracleScene(scene, trig_mode=pyoracle.TRIG_POLY, sky_bilinear=True).render(w, h, 8, threads=0)
    n_rgba, n_mean, _ = pyoracle.OracleScene(scene, trig_mode=pyoracle.TRIG_POLY).render(w, h, 8, threads=0)
    assert not np.array_equal(o_mean, n_mean)
    with rt.Renderer(0) as r:
        r.upload_scene(scene)
        r.set_option("sky_filter", "bilinear")
        for kernel in (("auto", "wave", "samples", "stream") if bvh else ("auto",)):
            r.set_option("tree_kernel", kernel)
            rgba, mean = r.render(w, h, seed=8, want_rgba8=True, want_mean=True)
            assert r.stats().segments == o_st.segments
            c = compare(mean, rgba, o_mean, o_rgba)
            assert c["mismatched_pixels"] == 0 and c["max_u8_delta"] == 0, (kernel, c)
        r.set_option("sky_filter", "nearest")
        r.set_option("tree_kernel", "auto")
        rgba, mean = r.render(w, h, seed=8, want_rgba8=True, want_mean=True)
        assert compare(mean, rgba, n_mean, n_rgba)["mismatched_pixels"] == 0
        with pytest.raises(rt.RtrError):
            r.set_option("sky_filter", "trilinear")


@pytest.mark.parametrize("builder,w,h", [(lambda: house_scene(3), 160, 90), (lambda: simple_scene(4), 96, 96)])
def test_traversal_counters_against_the_unculled_traversal(builder, w, h):
    """Option "traversal_counters": the frame does not change; the device-counted node steps and leaf tests are positive and
    bounded by the oracle's counts under the reference's unculled both-children traversal (bvh.rs:160-166: a node step
    tests both children's boxes, so 2 * node_steps <= AABB::hit calls; the oracle also counts the root's own box)."""
    scene = builder()
    with rt.Renderer(0) as r:
        r.upload_scene(scene)
        ref, _ = r.render(w, h, seed=21)
        plain = r.stats()
        assert plain.traversal_counted == 0 and plain.node_steps == 0
        r.set_option("traversal_counters", "on")
        got, _ = r.render(w, h, seed=21)
        st = r.stats()
        assert np.array_equal(ref, got) and st.segments == plain.segments
    _, _, o_st = pyoracle.OracleScene(scene).render(w, h, 21, threads=0)
    assert st.traversal_counted == 1
    assert 0 < 2 * st.node_steps <= o_st.box_tests
    assert 0 < st.leaf_tests <= o_st.sphere_tests + o_st.tri_tests
    assert st.node_steps >= st.segments * 0.5  # nearly every segment enters the root
    assert st.sky_lookups == o_st.sky_lookups


def _nested_fog_scene(bvh, mesh_boundary, spp=4):
    """A Volume whose boundary is a Volume (lib/volume.rs:20: `boundary: Box<dyn Object>`), next to ordinary objects."""
    from raytracingrust_amd.configs import mesh_path

    s = rt.Scene()
    s.camera = rt.Camera([0.0, 0.6, 3.5], [0.0, 0.0, 0.0], [0.0, 1.0, 0.0], 50.0, 16.0 / 9.0)
    s.background = rt.GradientBackground([1.0, 1.0, 1.0], [0.4, 0.6, 1.0])
    s.settings.samples_per_pixel = spp
    s.settings.enable_bvh_tree = bvh
    fog = s.add_material(rt.Isotropic([0.9, 0.85, 0.8]))
    ground = s.add_material(rt.Lambertian([0.4, 0.5, 0.4]))
    glass = s.add_material(rt.Dielectric(1.5))
    s.add_object(rt.Sphere([0.0, -101.0, 0.0], 100.0, ground))
    if mesh_boundary:
        inner = rt.Volume(rt.Mesh.from_file(mesh_path("cube.obj"), fog), 1.5)
    else:
        inner = rt.Volume(rt.Sphere([0.0, 0.0, 0.0], 1.0, fog), 1.5)
    s.add_object(rt.Volume(inner, 0.8))
    s.add_object(rt.Sphere([1.8, -0.4, 0.3], 0.6, glass))
    s.add_object(rt.Volume(rt.Sphere([-1.9, -0.3, 0.2], 0.7, fog), 2.0))  # an ordinary Volume beside it
    return s


@pytest.mark.parametrize("bvh", [True, False])
@pytest.mark.parametrize("mesh_boundary", [False, True])
def test_volume_over_a_volume(gpu_renderer, bvh, mesh_boundary):
    """Volume::hit whose two boundary.hit calls (volume.rs:35-36) are Volume::hit again: three random draws per visit, in
    call order; bit-identical to the oracle's recursive evaluation, tree and linear worlds, every tree kernel."""
    scene = _nested_fog_scene(bvh, mesh_boundary)
    c, st, o_st = run_both(gpu_renderer, scene, 160, 90, 11)
    assert st.segments == o_st.segments and o_st.rng_draws > 0
    assert c["mismatched_pixels"] == 0 and c["max_u8_delta"] == 0, c
    if bvh:
        ref, _ = gpu_renderer.render(160, 90, seed=11)
        for kernel in ("samples", "resumable", "wave", "stream"):
            with rt.Renderer(0) as r:
                r.set_option("tree_kernel", kernel)
                r.upload_scene(scene)
                got, _ = r.render(160, 90, seed=11)
                assert np.array_equal(got, ref), kernel
