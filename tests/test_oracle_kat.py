"""RNG-free known-answer tests that pin the CPU oracle to the reference's semantics.

The reference ships no tests or golden vectors, and cannot be built here, so every expected value below
is derived by hand from the cited reference lines; none depends on a random draw (or, where draws
happen, the expected value is the same for every draw)."""
import math

import numpy as np
import pytest

import pyoracle
import raytracingrust_amd as rt
from helpers import benchmark_scene, simple_scene

f32 = np.float32


def trunc_u8(x):
    """color.rs:34-41: (c*255.0) as u8, after render.rs:103-107's sqrt"""
    v = f32(np.sqrt(f32(x))) * f32(255.0)
    return int(min(max(math.floor(float(v)), 0), 255))


def render(scene, w, h, seed=1, **kw):
    return pyoracle.OracleScene(scene, **kw).render(w, h, seed, threads=0)


def float_sum_mean(value, n):
    """render.rs:84-107: color += sample (n times, f32), then / n"""
    acc = f32(0.0)
    for _ in range(n):
        acc = f32(acc + f32(value))
    return f32(acc / f32(n))


def test_empty_scene_uniform_background():
    s = rt.Scene()
    s.background = rt.UniformBackground([0.6, 0.6, 0.6])
    s.settings.samples_per_pixel = 7
    rgba, mean, st = render(s, 16, 9)
    assert np.all(mean == float_sum_mean(0.6, 7))
    assert np.all(rgba[..., :3] == 197) and np.all(rgba[..., 3] == 255)  # trunc(sqrt(0.6)*255) = 197
    assert trunc_u8(float_sum_mean(0.6, 7)) == 197
    assert st.segments == 16 * 9 * 7  # one world.hit per sample, all miss


def test_background_clamped_by_clamp_indirect():
    s = rt.Scene()
    s.background = rt.UniformBackground([4.0, 0.25, 100.0])
    s.settings.clamp_indirect = 2.0
    s.settings.samples_per_pixel = 3
    rgba, mean, _ = render(s, 4, 4)
    assert np.all(mean[..., 0] == 2.0) and np.all(mean[..., 1] == 0.25) and np.all(mean[..., 2] == 2.0)
    assert np.all(rgba[..., 0] == 255) and np.all(rgba[..., 1] == 127) and np.all(rgba[..., 2] == 255)


def test_max_ray_depth_zero_is_black():
    s = benchmark_scene(4, max_ray_depth=0)
    rgba, mean, st = render(s, 16, 12)
    assert np.all(mean == 0) and np.all(rgba[..., :3] == 0) and np.all(rgba[..., 3] == 255)
    assert st.segments == 0  # render.rs:34-36: depth test precedes the intersection


def test_emission_sphere_fills_view():
    s = rt.Scene()
    m = s.add_material(rt.Emission([1.0, 0.5, 0.25], 3.0))  # colour*strength = (3, 1.5, 0.75)
    s.add_object(rt.Sphere([0.0, 0.0, -2.0], 100.0, m))     # camera is inside: every ray hits
    s.settings.samples_per_pixel = 5
    s.settings.clamp_indirect = 2.0
    rgba, mean, st = render(s, 8, 8)
    assert np.all(mean[..., 0] == float_sum_mean(2.0, 5))
    assert np.all(mean[..., 1] == float_sum_mean(1.5, 5))
    assert np.all(mean[..., 2] == float_sum_mean(0.75, 5))
    assert np.all(rgba[..., 0] == 255) and np.all(rgba[..., 1] == 255) and np.all(rgba[..., 2] == trunc_u8(0.75))


def test_random_mode_is_a_silhouette():
    s = rt.Scene()
    s.background = rt.UniformBackground([0.6, 0.6, 0.6])
    m = s.add_material(rt.Lambertian([0.5, 0.5, 0.5]))
    s.add_object(rt.Sphere([0.0, 0.0, -1.0], 0.5, m))
    s.settings.mode = rt.RenderMode.Random
    s.settings.samples_per_pixel = 4
    rgba, mean, _ = render(s, 65, 65)
    assert np.all(rgba[32, 32, :3] == 0)   # centre: every sample hits -> black (render.rs:46-48)
    assert np.all(rgba[0, 0, :3] == 197)   # corner: every sample misses -> background
    vals = np.unique(mean)
    full = float_sum_mean(0.6, 4)
    assert vals.min() == 0.0 and vals.max() == full
    assert np.all(mean[28:37, 28:37] == 0)


def test_normal_mode_centre_pixel():
    s = rt.Scene()
    m = s.add_material(rt.Lambertian([0.5, 0.5, 0.5]))
    s.add_object(rt.Sphere([0.0, 0.0, -2.0], 1.0, m))
    s.settings.mode = rt.RenderMode.Normal
    s.settings.samples_per_pixel = 1
    o = pyoracle.OracleScene(s)
    h = o.probe_hit([0, 0, 0], [0, 0, -1])
    assert h["t"] == 1.0 and h["front_face"] and np.allclose(h["normal"], [0, 0, 1]) and np.allclose(h["point"], [0, 0, -1])
    rgba, mean, _ = o.render(33, 33, 1)
    # render.rs:42-45: 0.5*(n+1); at the image centre n ~ (0,0,1)
    assert np.allclose(mean[16, 16], [0.5, 0.5, 1.0], atol=0.03)


def test_white_furnace_dielectric():
    """Dielectric attenuation is (1,1,1) (materials.rs:134-137): every path that escapes carries the
    background radiance exactly; only depth cut-offs (black) can lower a pixel."""
    s = rt.Scene()
    s.background = rt.UniformBackground([0.6, 0.6, 0.6])
    g = s.add_material(rt.Dielectric(1.5))
    s.add_object(rt.Sphere([0.0, 0.0, -1.5], 0.7, g))
    s.add_object(rt.Sphere([0.9, 0.2, -1.2], 0.3, g))
    s.settings.samples_per_pixel = 6
    s.settings.max_ray_depth = 200
    rgba, mean, _ = render(s, 48, 48)
    full = float_sum_mean(0.6, 6)
    assert mean.max() == full
    assert np.count_nonzero(mean == full) >= 0.995 * mean.size
    assert np.all(mean <= full)


def test_single_lambertian_sphere_interior_is_albedo_times_background():
    """A convex Lambertian sphere alone under a uniform background: the scattered ray leaves the surface and
    escapes, so every interior sample = albedo*bg regardless of the random direction (materials.rs:31-41)."""
    s = rt.Scene()
    s.background = rt.UniformBackground([0.5, 0.5, 0.5])
    m = s.add_material(rt.Lambertian([0.8, 0.4, 0.2]))
    s.add_object(rt.Sphere([0.0, 0.0, -1.0], 0.5, m))
    s.settings.samples_per_pixel = 4
    rgba, mean, _ = render(s, 65, 65)
    inner = mean[26:39, 26:39]
    for c, a in enumerate([0.8, 0.4, 0.2]):
        expect = float_sum_mean(f32(a) * f32(0.5), 4)
        frac = np.count_nonzero(inner[..., c] == expect) / inner[..., c].size
        assert frac >= 0.98, (c, frac)  # grazing self-hits (tmin = 1e-5) may darken a stray sample
    assert tuple(rgba[32, 32, :3]) == (trunc_u8(0.4), trunc_u8(0.2), trunc_u8(0.1))


def test_depth_one_only_background_terms():
    """max_ray_depth = 1: a Lambertian hit scatters, the recursion returns black at depth 1 (render.rs:34-36),
    attenuation*black = 0; a miss returns the background."""
    s = rt.Scene()
    s.background = rt.UniformBackground([0.6, 0.6, 0.6])
    m = s.add_material(rt.Lambertian([0.5, 0.5, 0.5]))
    s.add_object(rt.Sphere([0.0, 0.0, -1.0], 0.5, m))
    s.settings.max_ray_depth = 1
    s.settings.samples_per_pixel = 2
    rgba, mean, _ = render(s, 33, 33)
    assert np.all(mean[16, 16] == 0) and np.all(mean[0, 0] == float_sum_mean(0.6, 2))


def test_camera_ray_origin_closed_form():
    """camera.rs:37-56 for the default camera (fov 90, aspect 1): h = tan(45 deg) = 1."""
    s = rt.Scene()
    ro = pyoracle.OracleScene(s).camera_ray_origin()
    h = f32(math.tan(float(f32(f32(90.0) * f32(math.pi) / f32(180.0)) / f32(2.0))))
    assert np.allclose(ro[0], [0, 0, 0])
    assert np.allclose(ro[1], [2 * h, 0, 0], rtol=1e-6)  # horizontal = viewport_width * u
    assert np.allclose(ro[2], [0, 2 * h, 0], rtol=1e-6)  # vertical
    assert np.allclose(ro[3], [-h, h, -1], rtol=1e-6)    # "lower_left_corner" is the UPPER left (camera.rs:54)
    s.camera = rt.Camera([-3, 2, 4], [0, 2, -1], [0, 1, 0], 90.0, 16 / 9)
    ro = pyoracle.OracleScene(s).camera_ray_origin()
    w = np.array([-3, 0, 5], dtype=np.float64) / math.sqrt(34.0)
    u = np.cross([0, 1, 0], w)
    u /= np.linalg.norm(u)
    v = np.cross(w, u)
    assert np.allclose(ro[1], 2 * 16 / 9 * u, rtol=1e-5, atol=1e-6)
    assert np.allclose(ro[2], 2 * v, rtol=1e-5, atol=1e-6)
    assert np.allclose(ro[3], np.array([-3, 2, 4]) - 16 / 9 * u + v - w, rtol=1e-5, atol=1e-6)


def test_aabb_slab_semantics():
    """aabb.rs:81-137: hit iff the narrowed interval stays non-empty (tmax <= tmin is a miss)."""
    inf = float("inf")
    box = ([-1, -1, -1], [1, 1, 1])
    assert pyoracle.probe_aabb(*box, [0, 0, 5], [0, 0, -1], 1e-5, inf)
    assert not pyoracle.probe_aabb(*box, [0, 0, 5], [0, 0, 1], 1e-5, inf)   # behind
    assert not pyoracle.probe_aabb(*box, [2, 0, 5], [0, 0, -1], 1e-5, inf)  # d.x = 0 outside the slab: +-inf bounds
    assert pyoracle.probe_aabb(*box, [0.5, 0.5, 5], [0, 0, -1], 1e-5, inf)  # d.x = d.y = 0 inside the slabs
    assert not pyoracle.probe_aabb(*box, [0, 0, 5], [0, 0, -1], 1e-5, 3.9)  # tmax before the entry (t = 4)
    assert pyoracle.probe_aabb(*box, [0, 0, 5], [0, 0, -1], 1e-5, 4.5)
    assert pyoracle.probe_aabb(*box, [0, 0, 0], [1, 1, 1], 1e-5, inf)       # origin inside
    # degenerate (zero-thickness) box: tmax <= tmin -> miss even for a ray through it
    assert not pyoracle.probe_aabb([-1, -1, 0], [1, 1, 0], [0, 0, 5], [0, 0, -1], 1e-5, inf)
    # exact corner graze: x interval [1,3] and y interval [3,5] share one point -> miss (<=)
    assert not pyoracle.probe_aabb([1, 3, -1], [3, 5, 1], [0, 0, 0], [1, 1, 0], 1e-5, inf)


def test_sphere_quadratic_closed_form():
    """objects.rs:22-51"""
    s = rt.Scene()
    s.settings.enable_bvh_tree = False
    m = s.add_material(rt.Lambertian([0.5, 0.5, 0.5]))
    s.add_object(rt.Sphere([0, 0, -3], 1.0, m))
    o = pyoracle.OracleScene(s)
    h = o.probe_hit([0, 0, 0], [0, 0, -2])  # unnormalised direction: t scales (half-b form, a = |d|^2)
    assert h["t"] == 1.0 and h["front_face"] and np.allclose(h["point"], [0, 0, -2]) and np.allclose(h["normal"], [0, 0, 1])
    h = o.probe_hit([0, 0, -3], [0, 0, -1])  # from the centre: first root negative -> far root, back face
    assert h["t"] == 1.0 and not h["front_face"] and np.allclose(h["normal"], [0, 0, 1])
    assert o.probe_hit([0, 2, 0], [0, 0, -1]) is None
    assert o.probe_hit([0, 0, 0], [0, 0, 1]) is None
    # negative radius flips the outward normal (objects.rs:46: /radius)
    s2 = rt.Scene()
    s2.add_material(rt.Lambertian([0.5, 0.5, 0.5]))
    s2.add_object(rt.Sphere([0, 0, -3], -1.0, 0))
    h = pyoracle.OracleScene(s2).probe_hit([0, 0, 0], [0, 0, -1])
    assert h["t"] == 2.0 and not h["front_face"]


def _one_triangle_scene(normals=None):
    s = rt.Scene()
    m = s.add_material(rt.Lambertian([0.5, 0.5, 0.5]))
    pos = np.array([[0, 0, -2], [1, 0, -2], [0, 1, -2]], dtype=np.float32)
    nrm = np.array([[0, 0, 1]] * 3, dtype=np.float32) if normals is None else np.asarray(normals, dtype=np.float32)
    s.add_object(rt.Mesh(pos, nrm, [0, 1, 2], m))
    return s


def test_triangle_moeller_trumbore_closed_form():
    """mesh.rs:146-197: t, face normal ((v2-v1)x(v0-v1) normalised = CCW normal), single-leaf mesh has no box."""
    o = pyoracle.OracleScene(_one_triangle_scene())
    h = o.probe_hit([0.25, 0.25, 0], [0, 0, -1])
    assert h["t"] == 2.0 and h["front_face"] and np.allclose(h["normal"], [0, 0, 1]) and np.allclose(h["point"], [0.25, 0.25, -2])
    assert o.probe_hit([0.75, 0.75, 0], [0, 0, -1]) is None  # u+v > 1
    assert o.probe_hit([-0.1, 0.2, 0], [0, 0, -1]) is None   # u < 0
    h = o.probe_hit([0.25, 0.25, -4], [0, 0, 1])             # from behind: normal flipped against the ray
    assert h["t"] == 2.0 and not h["front_face"] and np.allclose(h["normal"], [0, 0, -1])
    assert o.probe_hit([0.25, 0.25, 0], [1, 0, 0]) is None   # parallel: |a| < tmin (mesh.rs:166)
    # interpolated vertex normals differ from the face normal -> angle > 0 -> face normal wins (mesh.rs:187-189)
    o2 = pyoracle.OracleScene(_one_triangle_scene([[1, 0, 0], [0, 1, 0], [0, 0, 1]]))
    h = o2.probe_hit([0.25, 0.25, 0], [0, 0, -1])
    assert np.allclose(h["normal"], [0, 0, 1])


def test_bvh_builder_tree_shape_hand_placed():
    """bvh.rs:63-143: widest centroid axis with strict >, ties -> z; stable sort; left = first floor(n/2)."""

    def order(centers):
        s = rt.Scene()
        m = s.add_material(rt.Lambertian([0.5, 0.5, 0.5]))
        for c in centers:
            s.add_object(rt.Sphere(c, 0.1, m))
        return list(pyoracle.OracleScene(s).bvh_leaf_order())

    assert order([[3, 0, 0], [1, 0, 0], [2, 0, 0], [0, 0, 0]]) == [3, 1, 2, 0]  # x spread: sorted by x
    assert order([[0, 3, 0], [0, 1, 0], [0, 2, 0]]) == [1, 2, 0]                  # y spread
    assert order([[0, 0, 2], [1, 1, 0], [2, 2, 1]]) == [1, 2, 0]                  # spreads (2,2,2): tie -> z
    assert order([[0, 0, 0], [0, 0, 0], [0, 0, 0]]) == [0, 1, 2]                  # all equal: stable
    assert order([[1, 0, 0], [0, 5, 0], [1, 0, 1], [0, 5, 1], [1, 0, 2]]) == [0, 2, 4, 1, 3]  # y split, then z
    assert order([[4, 0, 0], [3, 0, 0], [2, 0, 0], [1, 0, 0], [0, 0, 0]]) == [4, 3, 2, 1, 0]


def test_bvh_and_linear_world_agree():
    """bvh.rs:160-166 vs scene.rs:141-157: same closest hit (no exact ties in this scene)."""
    ob = pyoracle.OracleScene(simple_scene(1, enable_bvh_tree=True))
    ol = pyoracle.OracleScene(simple_scene(1, enable_bvh_tree=False))
    rng = np.random.default_rng(5)
    hits = 0
    for _ in range(400):
        d = rng.normal(size=3).astype(np.float32)
        d[2] = -abs(d[2]) - 0.2
        o = np.array([rng.uniform(-0.5, 0.5), rng.uniform(-0.2, 0.3), 0.5], dtype=np.float32)
        a, b = ob.probe_hit(o, d), ol.probe_hit(o, d)
        assert (a is None) == (b is None)
        if a is not None:
            hits += 1
            assert a["t"] == b["t"] and a["material"] == b["material"] and np.array_equal(a["normal"], b["normal"])
    assert hits > 100


def test_rng_float_distributions():
    """rand 0.8 float conversions: 23-bit grids; the same k feeds both unit forms.  Kind 2 is
    the rejection sampler's LCG sub-stream (one word of the path's stream seeds it): gen_range(-1.0..1.0)'s grid."""
    u = pyoracle.probe_rng(42, 7, 3, 0, 4096)
    h = pyoracle.probe_rng(42, 7, 3, 1, 4096)
    g = pyoracle.probe_rng(42, 7, 3, 2, 4096)
    assert u.min() >= 0.0 and u.max() <= 1.0
    assert h.min() >= 0.0 and h.max() < 1.0
    assert g.min() >= -1.0 and g.max() < 1.0
    assert np.all(h * 2.0**23 == np.floor(h * 2.0**23))  # k * 2^-23
    assert np.all((g + 1.0) * 2.0**22 == np.floor((g + 1.0) * 2.0**22))
    assert np.array_equal(u, (h * f32(1.0 + 2.0**-23)).astype(np.float32))
    assert abs(float(h.mean()) - 0.5) < 0.03 and abs(float(g.mean())) < 0.05
    assert abs(float((g * g).mean()) - 1.0 / 3.0) < 0.03
    # the ball sampler takes ONE word of the path's stream per call: the draw after a Lambertian bounce does not
    # depend on how many candidates were rejected (probed through two streams that differ in the sample only)
    assert not np.array_equal(pyoracle.probe_rng(42, 7, 4, 2, 16), g[:16])
    # streams are keyed by (seed, pixel, sample)
    assert not np.array_equal(pyoracle.probe_rng(42, 7, 4, 1, 16), h[:16])
    assert not np.array_equal(pyoracle.probe_rng(42, 8, 3, 1, 16), h[:16])
    assert not np.array_equal(pyoracle.probe_rng(43, 7, 3, 1, 16), h[:16])


def test_polynomial_trig_matches_libm():
    """The HIP kernels' acos/atan2 (oracle mode OR_TRIG_POLY) against libm, which the reference calls."""
    L = pyoracle.lib()
    xs = np.linspace(-1, 1, 20001, dtype=np.float32)
    err = max(abs(L.or_probe_acos(float(x), 1) - L.or_probe_acos(float(x), 0)) for x in xs)
    assert err < 5e-7
    assert math.isnan(L.or_probe_acos(1.0000001, 1)) and math.isnan(L.or_probe_acos(1.0000001, 0))
    rng = np.random.default_rng(3)
    worst = 0.0
    for _ in range(20000):
        y, x = (float(f32(v)) for v in rng.normal(size=2))
        worst = max(worst, abs(L.or_probe_atan2(y, x, 1) - L.or_probe_atan2(y, x, 0)))
    assert worst < 5e-7
    for y, x in [(0.0, 1.0), (0.0, -1.0), (-0.0, -1.0), (1.0, 0.0), (-1.0, 0.0), (0.0, 0.0)]:
        assert L.or_probe_atan2(y, x, 1) == pytest.approx(L.or_probe_atan2(y, x, 0), abs=1e-7)


def test_skymap_lookup_orientation():
    """backgrounds.rs:84-92 + math.rs:24-29: theta = acos(-d.y), phi = atan2(-d.z, d.x) + pi;
    x = (phi/2pi * W) as usize % W; y = H - 1 - (theta/pi * H) as usize % H; nearest texel."""
    H, W = 8, 16
    img = np.zeros((H, W, 3), dtype=np.float32)
    img[..., 0] = np.arange(H, dtype=np.float32)[:, None]  # R = row
    img[..., 1] = np.arange(W, dtype=np.float32)[None, :]  # G = column
    s = rt.Scene()
    s.background = rt.SkyMap(img)
    s.settings.samples_per_pixel = 1
    s.settings.clamp_indirect = 100.0

    def look(d):
        s.camera = rt.Camera([0, 0, 0], d, [0, 1, 0] if abs(d[1]) < 0.9 else [0, 0, 1], 1.0, 1.0)
        _, mean, _ = pyoracle.OracleScene(s).render(3, 3, 1)
        return mean[1, 1]

    assert look([0, 1, 0])[0] == 0       # almost straight up: theta -> pi-, u -> 1-, (u*H) as usize = H-1 -> row 0 (top)
    assert look([0, -1, 0])[0] == H - 1  # almost straight down: u -> 0+ -> row H-1 (bottom)
    fwd = look([0, 0, -1])               # theta = pi/2 -> row H-1-H/2; phi = atan2(1,0)+pi = 1.5 pi -> v = .75
    # (the centre pixel's jittered ray points slightly right of and below the view axis)
    assert fwd[0] in (H - 1 - H // 2, H - H // 2) and fwd[1] in ((3 * W) // 4 - 1, (3 * W) // 4)
    assert look([1, 0, 0])[1] in (W // 2 - 1, W // 2)  # phi = atan2(-0, 1) + pi = pi -> v = .5


def test_polynomial_vs_libm_trig_on_a_skymap_frame():
    """The HIP kernels (and the oracle's parity mode) evaluate SkyMap's acos/atan2 with polynomials; the reference calls
    libm.  On a BASELINE-style HDRI frame the two differ only where a lookup sits on a texel border: far inside the
    1e-4 relative-L2 tolerance, RGBA8 identical."""
    from helpers import compare
    from raytracingrust_amd.configs import baseline_config

    scene, _, _ = baseline_config("c3l", 4)
    a = pyoracle.OracleScene(scene, trig_mode=pyoracle.TRIG_POLY).render(240, 135, 5, threads=0)
    b = pyoracle.OracleScene(scene, trig_mode=pyoracle.TRIG_LIBM).render(240, 135, 5, threads=0)
    assert a[2].sky_lookups == b[2].sky_lookups > 10000
    c = compare(a[1], a[0], b[1], b[0])
    assert max(c["rel_l2"]) < 1e-5 and c["max_u8_delta"] <= 1, c
    assert c["mismatched_pixels"] < 0.01 * 240 * 135


def test_volume_over_a_mesh_boundary_known_answer():
    """Volume::hit over a MESH boundary (volume.rs:34-73 with `boundary: Box<dyn Object>`), without depending on a random
    draw: with an enormous density, hit_distance = -1/density * ln(xi) is absorbed by t1, so the medium is hit exactly
    where the ray enters the boundary; the HitRecord's normal is the constant (1, 0, 0) (volume.rs:67), face-forwarded by
    HitRecord::new (ray.rs:39-54: flipped when d.x > 0), which RenderMode::Normal turns into the colour 0.5 * (n + 1) at
    depth 0 (render.rs:42-45): (1, 0.5, 0.5) left of the frame's centre, (0, 0.5, 0.5) right of it; sqrt -> RGBA8
    (255, 180, 180) / (0, 180, 180) on every pixel whose samples all enter the cube, the background elsewhere."""
    from raytracingrust_amd.configs import mesh_path

    s = rt.Scene()
    s.camera = rt.Camera([0.0, 0.0, 4.0], [0.0, 0.0, 0.0], [0.0, 1.0, 0.0], 60.0, 1.0)
    s.settings.samples_per_pixel = 4
    s.settings.mode = rt.RenderMode.Normal
    s.background = rt.UniformBackground([0.0, 0.0, 0.0])
    m = s.add_material(rt.Isotropic([1.0, 1.0, 1.0]))
    cube = rt.Mesh.from_file(mesh_path("cube.obj"), m)
    lo, hi = cube.positions.min(axis=0), cube.positions.max(axis=0)
    assert np.allclose(lo, -1.0) and np.allclose(hi, 1.0)  # the asset is the cube [-1, 1]^3
    s.add_object(rt.Volume(cube, 1.0e30))
    for bvh in (True, False):
        s.settings.enable_bvh_tree = bvh
        rgba, mean, st = pyoracle.OracleScene(s).render(64, 64, 5, threads=0)
        assert st.segments == 64 * 64 * 4  # Normal mode returns at depth 0
        left = rgba[24:40, 24:31, :3].reshape(-1, 3)  # the cube's front face covers the centre of the frame; d.x < 0 here
        right = rgba[24:40, 33:40, :3].reshape(-1, 3)  # d.x > 0: the constant normal is flipped to (-1, 0, 0)
        assert np.all(left == np.array([255, 180, 180], dtype=np.uint8)), np.unique(left, axis=0)
        assert np.all(right == np.array([0, 180, 180], dtype=np.uint8)), np.unique(right, axis=0)
        assert np.all(rgba[0, :, :3] == 0) and np.all(rgba[:, 0, :3] == 0)  # the corners see the background


def test_bilinear_sky_filter_known_answers():
    """The oracle's restatement of the opt-in bilinear SkyMap filter: on a constant image it is the nearest-texel frame
    (the weights sum to one: a * (1 - t) + a * t == a only up to rounding, so compare within 2 ulp); on an image that is
    linear in the column index it reproduces the column coordinate between texel centres."""
    s = rt.Scene()
    s.settings.samples_per_pixel = 1
    s.camera = rt.Camera([0.0, 0.0, 0.0], [0.0, 0.0, -1.0], [0.0, 1.0, 0.0], 90.0, 1.0)
    s.background = rt.SkyMap(np.full((8, 16, 3), 0.37, dtype=np.float32))
    _, a, _ = pyoracle.OracleScene(s, sky_bilinear=True).render(32, 32, 3, threads=0)
    _, b, _ = pyoracle.OracleScene(s).render(32, 32, 3, threads=0)
    assert np.all(b == np.float32(0.37)) and np.max(np.abs(a - b)) <= 2 * np.spacing(np.float32(0.37))
    ramp = np.zeros((4, 64, 3), dtype=np.float32)
    ramp[:, :, 0] = (np.arange(64, dtype=np.float32) / np.float32(64.0))[None, :]  # (stays below clamp_indirect)
    s.background = rt.SkyMap(ramp)
    _, lin, _ = pyoracle.OracleScene(s, sky_bilinear=True).render(64, 8, 3, threads=0)
    _, near, _ = pyoracle.OracleScene(s).render(64, 8, 3, threads=0)
    red_l, red_n = lin[4, :, 0] * np.float32(64.0), near[4, :, 0] * np.float32(64.0)
    assert np.all(red_n == np.floor(red_n))  # nearest: whole column indices
    inner = (red_n > 1) & (red_n < 62)       # away from the longitude seam the interpolated value lies within half a texel
    assert np.all(np.abs(red_l[inner] - red_n[inner]) <= 0.5 + 1e-4) and np.any(red_l[inner] != np.floor(red_l[inner]))
