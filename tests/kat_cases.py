"""RNG-free known-answer cases of SURVEY.md section 4, written ONCE against a small backend interface so that the same
closed-form expectations are asserted

  * on the CPU oracle (tests/test_oracle_kat.py: pins the restatement to the reference's semantics), and
  * on the HIP path through the C ABI (tests/test_gpu_kat.py, `-m gpu`: the only evidence about the product that does
    not route through oracle/ -- the reference ships no fixtures and cannot be built here).

Every expected value is derived by hand from the cited reference lines; none depends on a random draw (or, where draws
happen, the expected value is the same for every draw).  A backend offers

    render(scene, w, h, seed=1, sky_bilinear=False) -> (rgba8[h,w,4], mean[h,w,3] float32, segments)
    probe_hit(scene, origin, direction)            -> dict(t, point, normal, front_face, material) | None
                                                      (world.hit(ray, 0.00001, inf), render.rs:38)
    camera_ray_origin(scene)                       -> 4 x 3 (origin, horizontal, vertical, lower_left_corner)
    leaf_order(scene)                              -> object indices of the scene tree's leaves, depth first
"""
import math

import numpy as np

import raytracingrust_amd as rt
from raytracingrust_amd.configs import mesh_path, simple_scene, benchmark_scene

f32 = np.float32


def trunc_u8(x):
    """color.rs:34-41: (c*255.0) as u8, after render.rs:103-107's sqrt"""
    v = f32(np.sqrt(f32(x))) * f32(255.0)
    return int(min(max(math.floor(float(v)), 0), 255))


def float_sum_mean(value, n):
    """render.rs:84-107: color += sample (n times, f32), then / n"""
    acc = f32(0.0)
    for _ in range(n):
        acc = f32(acc + f32(value))
    return f32(acc / f32(n))


# --- frames ------------------------------------------------------------------------------------------------------------

def kat_empty_scene_uniform_background(b):
    s = rt.Scene()
    s.background = rt.UniformBackground([0.6, 0.6, 0.6])
    s.settings.samples_per_pixel = 7
    for bvh in (True, False):
        s.settings.enable_bvh_tree = bvh
        rgba, mean, segments = b.render(s, 16, 9)
        assert np.all(mean == float_sum_mean(0.6, 7))
        assert np.all(rgba[..., :3] == 197) and np.all(rgba[..., 3] == 255)  # trunc(sqrt(0.6)*255) = 197
        assert segments == 16 * 9 * 7  # one world.hit per sample, all miss
    assert trunc_u8(float_sum_mean(0.6, 7)) == 197


def kat_background_clamped_by_clamp_indirect(b):
    s = rt.Scene()
    s.background = rt.UniformBackground([4.0, 0.25, 100.0])
    s.settings.clamp_indirect = 2.0
    s.settings.samples_per_pixel = 3
    rgba, mean, _ = b.render(s, 4, 4)
    assert np.all(mean[..., 0] == 2.0) and np.all(mean[..., 1] == 0.25) and np.all(mean[..., 2] == 2.0)
    assert np.all(rgba[..., 0] == 255) and np.all(rgba[..., 1] == 127) and np.all(rgba[..., 2] == 255)


def kat_max_ray_depth_zero_is_black(b):
    for s in (benchmark_scene(4, max_ray_depth=0), simple_scene(2, max_ray_depth=0)):
        rgba, mean, segments = b.render(s, 16, 12)
        assert np.all(mean == 0) and np.all(rgba[..., :3] == 0) and np.all(rgba[..., 3] == 255)
        assert segments == 0  # render.rs:34-36: depth test precedes the intersection


def kat_emission_sphere_fills_view(b):
    s = rt.Scene()
    m = s.add_material(rt.Emission([1.0, 0.5, 0.25], 3.0))  # colour*strength = (3, 1.5, 0.75)
    s.add_object(rt.Sphere([0.0, 0.0, -2.0], 100.0, m))     # camera is inside: every ray hits
    s.settings.samples_per_pixel = 5
    s.settings.clamp_indirect = 2.0
    for bvh in (True, False):
        s.settings.enable_bvh_tree = bvh
        rgba, mean, segments = b.render(s, 8, 8)
        assert segments == 8 * 8 * 5  # Emission::scatter returns no ray (materials.rs:86-91)
        assert np.all(mean[..., 0] == float_sum_mean(2.0, 5))
        assert np.all(mean[..., 1] == float_sum_mean(1.5, 5))
        assert np.all(mean[..., 2] == float_sum_mean(0.75, 5))
        assert np.all(rgba[..., 0] == 255) and np.all(rgba[..., 1] == 255) and np.all(rgba[..., 2] == trunc_u8(0.75))


def kat_random_mode_is_a_silhouette(b):
    s = rt.Scene()
    s.background = rt.UniformBackground([0.6, 0.6, 0.6])
    m = s.add_material(rt.Lambertian([0.5, 0.5, 0.5]))
    s.add_object(rt.Sphere([0.0, 0.0, -1.0], 0.5, m))
    s.settings.mode = rt.RenderMode.Random
    s.settings.samples_per_pixel = 4
    rgba, mean, segments = b.render(s, 65, 65)
    assert segments == 65 * 65 * 4          # the mode returns at depth 0 (render.rs:46-48)
    assert np.all(rgba[32, 32, :3] == 0)    # centre: every sample hits -> black
    assert np.all(rgba[0, 0, :3] == 197)    # corner: every sample misses -> background
    vals = np.unique(mean)
    full = float_sum_mean(0.6, 4)
    assert vals.min() == 0.0 and vals.max() == full
    assert np.all(mean[28:37, 28:37] == 0)


def kat_normal_mode_centre_pixel(b):
    s = rt.Scene()
    m = s.add_material(rt.Lambertian([0.5, 0.5, 0.5]))
    s.add_object(rt.Sphere([0.0, 0.0, -2.0], 1.0, m))
    s.settings.mode = rt.RenderMode.Normal
    s.settings.samples_per_pixel = 1
    h = b.probe_hit(s, [0, 0, 0], [0, 0, -1])
    assert h["t"] == 1.0 and h["front_face"] and np.allclose(h["normal"], [0, 0, 1]) and np.allclose(h["point"], [0, 0, -1])
    _, mean, _ = b.render(s, 33, 33)
    # render.rs:42-45: 0.5*(n+1); at the image centre n ~ (0,0,1)
    assert np.allclose(mean[16, 16], [0.5, 0.5, 1.0], atol=0.03)
    # the sphere's silhouette: left of centre the outward normal leans to -x, right of centre to +x
    assert mean[16, 6, 0] < 0.3 and mean[16, 26, 0] > 0.7


def kat_white_furnace_dielectric(b):
    """Dielectric attenuation is (1,1,1) (materials.rs:134-137): every path that escapes carries the
    background radiance exactly; only depth cut-offs (black) can lower a pixel."""
    s = rt.Scene()
    s.background = rt.UniformBackground([0.6, 0.6, 0.6])
    g = s.add_material(rt.Dielectric(1.5))
    s.add_object(rt.Sphere([0.0, 0.0, -1.5], 0.7, g))
    s.add_object(rt.Sphere([0.9, 0.2, -1.2], 0.3, g))
    s.settings.samples_per_pixel = 6
    s.settings.max_ray_depth = 200
    for bvh in (True, False):
        s.settings.enable_bvh_tree = bvh
        _, mean, _ = b.render(s, 48, 48)
        full = float_sum_mean(0.6, 6)
        assert mean.max() == full
        assert np.count_nonzero(mean == full) >= 0.995 * mean.size
        assert np.all(mean <= full)


def _one_sphere_scene(material, bg=0.5):
    s = rt.Scene()
    s.background = rt.UniformBackground([bg, bg, bg])
    m = s.add_material(material)
    s.add_object(rt.Sphere([0.0, 0.0, -1.0], 0.5, m))
    s.settings.samples_per_pixel = 4
    return s


def kat_single_lambertian_sphere_interior_is_albedo_times_background(b):
    """A convex Lambertian sphere alone under a uniform background: the scattered ray leaves the surface and
    escapes, so every interior sample = albedo*bg regardless of the random direction (materials.rs:31-41)."""
    s = _one_sphere_scene(rt.Lambertian([0.8, 0.4, 0.2]))
    for bvh in (True, False):
        s.settings.enable_bvh_tree = bvh
        rgba, mean, _ = b.render(s, 65, 65)
        inner = mean[26:39, 26:39]
        for c, a in enumerate([0.8, 0.4, 0.2]):
            expect = float_sum_mean(f32(a) * f32(0.5), 4)
            frac = np.count_nonzero(inner[..., c] == expect) / inner[..., c].size
            assert frac >= 0.98, (c, frac)  # grazing self-hits (tmin = 1e-5) may darken a stray sample
        assert tuple(rgba[32, 32, :3]) == (trunc_u8(0.4), trunc_u8(0.2), trunc_u8(0.1))


def kat_single_mirror_sphere_interior_is_albedo_times_background(b):
    """Metal with fuzz 0 (materials.rs:58-70): reflect(d, n) about the front-face normal always has a positive component
    along n, so the bounce is kept and escapes a lone convex sphere: every interior sample = albedo*bg, whatever the
    (still drawn) random unit vector was -- it is scaled by fuzz = 0."""
    s = _one_sphere_scene(rt.Metal([0.9, 0.6, 0.3], 0.0))
    for bvh in (True, False):
        s.settings.enable_bvh_tree = bvh
        rgba, mean, segments = b.render(s, 65, 65)
        inner = mean[26:39, 26:39]
        for c, a in enumerate([0.9, 0.6, 0.3]):
            expect = float_sum_mean(f32(a) * f32(0.5), 4)
            frac = np.count_nonzero(inner[..., c] == expect) / inner[..., c].size
            assert frac >= 0.98, (c, frac)
        assert tuple(rgba[32, 32, :3]) == (trunc_u8(0.45), trunc_u8(0.3), trunc_u8(0.15))
        assert segments >= 65 * 65 * 4


def kat_depth_one_only_background_terms(b):
    """max_ray_depth = 1: a Lambertian hit scatters, the recursion returns black at depth 1 (render.rs:34-36),
    attenuation*black = 0; a miss returns the background."""
    s = rt.Scene()
    s.background = rt.UniformBackground([0.6, 0.6, 0.6])
    m = s.add_material(rt.Lambertian([0.5, 0.5, 0.5]))
    s.add_object(rt.Sphere([0.0, 0.0, -1.0], 0.5, m))
    s.settings.max_ray_depth = 1
    s.settings.samples_per_pixel = 2
    rgba, mean, segments = b.render(s, 33, 33)
    assert segments == 33 * 33 * 2  # one world.hit per sample: the second call returns before it (render.rs:34-36)
    assert np.all(mean[16, 16] == 0) and np.all(mean[0, 0] == float_sum_mean(0.6, 2))


def kat_skymap_lookup_orientation(b):
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
        _, mean, _ = b.render(s, 3, 3)
        return mean[1, 1]

    assert look([0, 1, 0])[0] == 0       # almost straight up: theta -> pi-, u -> 1-, (u*H) as usize = H-1 -> row 0 (top)
    assert look([0, -1, 0])[0] == H - 1  # almost straight down: u -> 0+ -> row H-1 (bottom)
    fwd = look([0, 0, -1])               # theta = pi/2 -> row H-1-H/2; phi = atan2(1,0)+pi = 1.5 pi -> v = .75
    # (the centre pixel's jittered ray points slightly right of and below the view axis)
    assert fwd[0] in (H - 1 - H // 2, H - H // 2) and fwd[1] in ((3 * W) // 4 - 1, (3 * W) // 4)
    assert look([1, 0, 0])[1] in (W // 2 - 1, W // 2)  # phi = atan2(-0, 1) + pi = pi -> v = .5


def kat_gradient_background_endpoints(b):
    """backgrounds.rs:44-49: t = 0.5 * (unit(d).y + 1); top * (1 - t) + bottom * t.  Looking (almost) straight up t -> 1 and
    the colour -> `bottom`; straight down t -> 0 and the colour -> `top` (the names are the reference's)."""
    s = rt.Scene()
    s.background = rt.GradientBackground([1.0, 0.0, 0.25], [0.0, 1.0, 0.75])
    s.settings.samples_per_pixel = 1

    def look(d):
        s.camera = rt.Camera([0, 0, 0], d, [0, 0, 1], 1.0, 1.0)
        _, mean, _ = b.render(s, 3, 3)
        return mean[1, 1]

    up, down = look([0, 1, 0]), look([0, -1, 0])
    assert np.allclose(up, [0.0, 1.0, 0.75], atol=2e-4) and np.allclose(down, [1.0, 0.0, 0.25], atol=2e-4)
    assert up[1] <= 1.0 and down[0] <= 1.0


def kat_volume_over_a_mesh_boundary(b):
    """Volume::hit over a MESH boundary (volume.rs:34-73 with `boundary: Box<dyn Object>`), without depending on a random
    draw: with an enormous density, hit_distance = -1/density * ln(xi) is absorbed by t1, so the medium is hit exactly
    where the ray enters the boundary; the HitRecord's normal is the constant (1, 0, 0) (volume.rs:67), face-forwarded by
    HitRecord::new (ray.rs:39-54: flipped when d.x > 0), which RenderMode::Normal turns into the colour 0.5 * (n + 1) at
    depth 0 (render.rs:42-45): (1, 0.5, 0.5) left of the frame's centre, (0, 0.5, 0.5) right of it; sqrt -> RGBA8
    (255, 180, 180) / (0, 180, 180) on every pixel whose samples all enter the cube, the background elsewhere."""
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
        rgba, mean, segments = b.render(s, 64, 64, seed=5)
        assert segments == 64 * 64 * 4  # Normal mode returns at depth 0
        left = rgba[24:40, 24:31, :3].reshape(-1, 3)  # the cube's front face covers the centre of the frame; d.x < 0 here
        right = rgba[24:40, 33:40, :3].reshape(-1, 3)  # d.x > 0: the constant normal is flipped to (-1, 0, 0)
        assert np.all(left == np.array([255, 180, 180], dtype=np.uint8)), np.unique(left, axis=0)
        assert np.all(right == np.array([0, 180, 180], dtype=np.uint8)), np.unique(right, axis=0)
        assert np.all(rgba[0, :, :3] == 0) and np.all(rgba[:, 0, :3] == 0)  # the corners see the background


def kat_bilinear_sky_filter(b):
    """The opt-in bilinear SkyMap filter (an extension): on a constant image it is the nearest-texel frame (the weights sum
    to one: a * (1 - t) + a * t == a only up to rounding, so compare within 2 ulp); on an image that is linear in the column
    index it reproduces the column coordinate between texel centres."""
    s = rt.Scene()
    s.settings.samples_per_pixel = 1
    s.camera = rt.Camera([0.0, 0.0, 0.0], [0.0, 0.0, -1.0], [0.0, 1.0, 0.0], 90.0, 1.0)
    s.background = rt.SkyMap(np.full((8, 16, 3), 0.37, dtype=np.float32))
    _, a, _ = b.render(s, 32, 32, seed=3, sky_bilinear=True)
    _, n, _ = b.render(s, 32, 32, seed=3)
    assert np.all(n == np.float32(0.37)) and np.max(np.abs(a - n)) <= 2 * np.spacing(np.float32(0.37))
    ramp = np.zeros((4, 64, 3), dtype=np.float32)
    ramp[:, :, 0] = (np.arange(64, dtype=np.float32) / np.float32(64.0))[None, :]  # (stays below clamp_indirect)
    s.background = rt.SkyMap(ramp)
    _, lin, _ = b.render(s, 64, 8, seed=3, sky_bilinear=True)
    _, near, _ = b.render(s, 64, 8, seed=3)
    red_l, red_n = lin[4, :, 0] * np.float32(64.0), near[4, :, 0] * np.float32(64.0)
    assert np.all(red_n == np.floor(red_n))  # nearest: whole column indices
    inner = (red_n > 1) & (red_n < 62)       # away from the longitude seam the interpolated value lies within half a texel
    assert np.all(np.abs(red_l[inner] - red_n[inner]) <= 0.5 + 1e-4) and np.any(red_l[inner] != np.floor(red_l[inner]))


# --- single rays -------------------------------------------------------------------------------------------------------

def kat_sphere_quadratic_closed_form(b):
    """objects.rs:22-51"""
    for bvh in (False, True):
        s = rt.Scene()
        s.settings.enable_bvh_tree = bvh
        m = s.add_material(rt.Lambertian([0.5, 0.5, 0.5]))
        s.add_object(rt.Sphere([0, 0, -3], 1.0, m))
        h = b.probe_hit(s, [0, 0, 0], [0, 0, -2])  # unnormalised direction: t scales (half-b form, a = |d|^2)
        assert h["t"] == 1.0 and h["front_face"] and np.allclose(h["point"], [0, 0, -2]) and np.allclose(h["normal"], [0, 0, 1])
        assert h["material"] == m.index
        h = b.probe_hit(s, [0, 0, -3], [0, 0, -1])  # from the centre: first root negative -> far root, back face
        assert h["t"] == 1.0 and not h["front_face"] and np.allclose(h["normal"], [0, 0, 1])
        assert b.probe_hit(s, [0, 2, 0], [0, 0, -1]) is None
        assert b.probe_hit(s, [0, 0, 0], [0, 0, 1]) is None
        # negative radius flips the outward normal (objects.rs:46: /radius)
        s2 = rt.Scene()
        s2.settings.enable_bvh_tree = bvh
        s2.add_material(rt.Lambertian([0.5, 0.5, 0.5]))
        s2.add_object(rt.Sphere([0, 0, -3], -1.0, 0))
        h = b.probe_hit(s2, [0, 0, 0], [0, 0, -1])
        assert h["t"] == 2.0 and not h["front_face"]


def _one_triangle_scene(normals=None, bvh=True):
    s = rt.Scene()
    s.settings.enable_bvh_tree = bvh
    m = s.add_material(rt.Lambertian([0.5, 0.5, 0.5]))
    pos = np.array([[0, 0, -2], [1, 0, -2], [0, 1, -2]], dtype=np.float32)
    nrm = np.array([[0, 0, 1]] * 3, dtype=np.float32) if normals is None else np.asarray(normals, dtype=np.float32)
    s.add_object(rt.Mesh(pos, nrm, [0, 1, 2], m))
    return s


def kat_triangle_moeller_trumbore_closed_form(b):
    """mesh.rs:146-197: t, face normal ((v2-v1)x(v0-v1) normalised = CCW normal), single-leaf mesh has no box."""
    for bvh in (True, False):
        s = _one_triangle_scene(bvh=bvh)
        h = b.probe_hit(s, [0.25, 0.25, 0], [0, 0, -1])
        assert h["t"] == 2.0 and h["front_face"] and np.allclose(h["normal"], [0, 0, 1]) and np.allclose(h["point"], [0.25, 0.25, -2])
        assert b.probe_hit(s, [0.75, 0.75, 0], [0, 0, -1]) is None  # u+v > 1
        assert b.probe_hit(s, [-0.1, 0.2, 0], [0, 0, -1]) is None   # u < 0
        h = b.probe_hit(s, [0.25, 0.25, -4], [0, 0, 1])             # from behind: normal flipped against the ray
        assert h["t"] == 2.0 and not h["front_face"] and np.allclose(h["normal"], [0, 0, -1])
        assert b.probe_hit(s, [0.25, 0.25, 0], [1, 0, 0]) is None   # parallel: |a| < tmin (mesh.rs:166)
        # interpolated vertex normals differ from the face normal -> angle > 0 -> face normal wins (mesh.rs:187-189)
        s2 = _one_triangle_scene([[1, 0, 0], [0, 1, 0], [0, 0, 1]], bvh=bvh)
        h = b.probe_hit(s2, [0.25, 0.25, 0], [0, 0, -1])
        assert np.allclose(h["normal"], [0, 0, 1])


def kat_boxed_mesh_hits_and_misses(b):
    """A two-triangle mesh has a Split root, so rays reach the triangles through AABB::hit (aabb.rs:81-137) on the mesh's
    box -- seeded from the zero box, i.e. always containing the origin (mesh.rs:200-213) -- and on the per-triangle boxes
    (flat triangles expanded to +-0.01 around the centroid on the thin axis, mesh.rs:203-212)."""
    for bvh in (True, False):
        s = rt.Scene()
        s.settings.enable_bvh_tree = bvh
        m = s.add_material(rt.Lambertian([0.5, 0.5, 0.5]))
        pos = np.array([[0, 0, -2], [1, 0, -2], [0, 1, -2], [1, 1, -2]], dtype=np.float32)
        nrm = np.array([[0, 0, 1]] * 4, dtype=np.float32)
        s.add_object(rt.Mesh(pos, nrm, [0, 1, 2, 1, 3, 2], m))
        h = b.probe_hit(s, [0.25, 0.25, 0], [0, 0, -1])
        assert h is not None and h["t"] == 2.0 and np.allclose(h["normal"], [0, 0, 1])
        h = b.probe_hit(s, [0.75, 0.75, 1], [0, 0, -1])  # the second triangle
        assert h is not None and h["t"] == 3.0 and np.allclose(h["point"], [0.75, 0.75, -2])
        assert b.probe_hit(s, [1.5, 0.5, 0], [0, 0, -1]) is None    # beside the quad
        assert b.probe_hit(s, [0.25, 0.25, 0], [0, 0, 1]) is None   # pointing away
        h = b.probe_hit(s, [0.5, 0.5, -5], [0, 0, 3])               # from behind, unnormalised: t = 3 / 3
        assert h is not None and h["t"] == 1.0 and not h["front_face"]


def kat_bvh_and_linear_world_agree(b):
    """bvh.rs:160-166 vs scene.rs:141-157: same closest hit (no exact ties in this scene)."""
    sb, sl = simple_scene(1, enable_bvh_tree=True), simple_scene(1, enable_bvh_tree=False)
    rng = np.random.default_rng(5)
    origins, dirs = [], []
    for _ in range(400):
        d = rng.normal(size=3).astype(np.float32)
        d[2] = -abs(d[2]) - 0.2
        origins.append(np.array([rng.uniform(-0.5, 0.5), rng.uniform(-0.2, 0.3), 0.5], dtype=np.float32))
        dirs.append(d)
    ha, hb = b.probe_hits(sb, origins, dirs), b.probe_hits(sl, origins, dirs)
    hits = 0
    for x, y in zip(ha, hb):
        assert (x is None) == (y is None)
        if x is not None:
            hits += 1
            assert x["t"] == y["t"] and x["material"] == y["material"] and np.array_equal(x["normal"], y["normal"])
    assert hits > 100


# --- host arithmetic ---------------------------------------------------------------------------------------------------

def kat_camera_ray_origin_closed_form(b):
    """camera.rs:37-56 for the default camera (fov 90, aspect 1): h = tan(45 deg) = 1."""
    s = rt.Scene()
    ro = b.camera_ray_origin(s)
    h = f32(math.tan(float(f32(f32(90.0) * f32(math.pi) / f32(180.0)) / f32(2.0))))
    assert np.allclose(ro[0], [0, 0, 0])
    assert np.allclose(ro[1], [2 * h, 0, 0], rtol=1e-6)  # horizontal = viewport_width * u
    assert np.allclose(ro[2], [0, 2 * h, 0], rtol=1e-6)  # vertical
    assert np.allclose(ro[3], [-h, h, -1], rtol=1e-6)    # "lower_left_corner" is the UPPER left (camera.rs:54)
    s.camera = rt.Camera([-3, 2, 4], [0, 2, -1], [0, 1, 0], 90.0, 16 / 9)
    ro = b.camera_ray_origin(s)
    w = np.array([-3, 0, 5], dtype=np.float64) / math.sqrt(34.0)
    u = np.cross([0, 1, 0], w)
    u /= np.linalg.norm(u)
    v = np.cross(w, u)
    assert np.allclose(ro[1], 2 * 16 / 9 * u, rtol=1e-5, atol=1e-6)
    assert np.allclose(ro[2], 2 * v, rtol=1e-5, atol=1e-6)
    assert np.allclose(ro[3], np.array([-3, 2, 4]) - 16 / 9 * u + v - w, rtol=1e-5, atol=1e-6)


def kat_bvh_builder_tree_shape_hand_placed(b):
    """bvh.rs:63-143: widest centroid axis with strict >, ties -> z; stable sort; left = first floor(n/2)."""

    def order(centers):
        s = rt.Scene()
        m = s.add_material(rt.Lambertian([0.5, 0.5, 0.5]))
        for c in centers:
            s.add_object(rt.Sphere(c, 0.1, m))
        return [int(i) for i in b.leaf_order(s)]

    assert order([[3, 0, 0], [1, 0, 0], [2, 0, 0], [0, 0, 0]]) == [3, 1, 2, 0]  # x spread: sorted by x
    assert order([[0, 3, 0], [0, 1, 0], [0, 2, 0]]) == [1, 2, 0]                  # y spread
    assert order([[0, 0, 2], [1, 1, 0], [2, 2, 1]]) == [1, 2, 0]                  # spreads (2,2,2): tie -> z
    assert order([[0, 0, 0], [0, 0, 0], [0, 0, 0]]) == [0, 1, 2]                  # all equal: stable
    assert order([[1, 0, 0], [0, 5, 0], [1, 0, 1], [0, 5, 1], [1, 0, 2]]) == [0, 2, 4, 1, 3]  # y split, then z
    assert order([[4, 0, 0], [3, 0, 0], [2, 0, 0], [1, 0, 0], [0, 0, 0]]) == [4, 3, 2, 1, 0]


FRAME_CASES = [kat_empty_scene_uniform_background, kat_background_clamped_by_clamp_indirect, kat_max_ray_depth_zero_is_black,
               kat_emission_sphere_fills_view, kat_random_mode_is_a_silhouette, kat_normal_mode_centre_pixel,
               kat_white_furnace_dielectric, kat_single_lambertian_sphere_interior_is_albedo_times_background,
               kat_single_mirror_sphere_interior_is_albedo_times_background, kat_depth_one_only_background_terms,
               kat_skymap_lookup_orientation, kat_gradient_background_endpoints, kat_volume_over_a_mesh_boundary,
               kat_bilinear_sky_filter]
RAY_CASES = [kat_sphere_quadratic_closed_form, kat_triangle_moeller_trumbore_closed_form, kat_boxed_mesh_hits_and_misses,
             kat_bvh_and_linear_world_agree]
HOST_CASES = [kat_camera_ray_origin_closed_form, kat_bvh_builder_tree_shape_hand_placed]


# --- backends ----------------------------------------------------------------------------------------------------------

class OracleBackend:
    """oracle/rt_oracle.c (the CPU restatement)."""
    name = "oracle"

    def render(self, scene, w, h, seed=1, sky_bilinear=False):
        import pyoracle

        rgba, mean, st = pyoracle.OracleScene(scene, sky_bilinear=sky_bilinear).render(w, h, seed, threads=0)
        return rgba, mean, int(st.segments)

    def probe_hit(self, scene, origin, direction):
        import pyoracle

        return pyoracle.OracleScene(scene).probe_hit(origin, direction)

    def probe_hits(self, scene, origins, directions):
        import pyoracle

        o = pyoracle.OracleScene(scene)
        return [o.probe_hit(a, d) for a, d in zip(origins, directions)]

    def camera_ray_origin(self, scene):
        import pyoracle

        return pyoracle.OracleScene(scene).camera_ray_origin()

    def leaf_order(self, scene):
        import pyoracle

        return list(pyoracle.OracleScene(scene).bvh_leaf_order())


class HipBackend:
    """The product: librtr_hip.so through its C ABI (ctypes).  `device` = -1 serves the host-arithmetic cases only."""
    name = "hip"

    def __init__(self, device=0):
        self.r = rt.Renderer(device)

    def close(self):
        self.r.close()

    def render(self, scene, w, h, seed=1, sky_bilinear=False):
        self.r.set_option("sky_filter", "bilinear" if sky_bilinear else "nearest")
        self.r.upload_scene(scene)
        rgba, mean = self.r.render(w, h, seed=seed, want_rgba8=True, want_mean=True)
        self.r.set_option("sky_filter", "nearest")
        return rgba, mean, int(self.r.stats().segments)

    def probe_hits(self, scene, origins, directions):
        self.r.upload_scene(scene)
        return self.r.probe_hits(origins, directions, enable_bvh_tree=bool(scene.settings.enable_bvh_tree))

    def probe_hit(self, scene, origin, direction):
        return self.probe_hits(scene, [origin], [direction])[0]

    def camera_ray_origin(self, scene):
        from raytracingrust_amd.render import camera_ray_origin

        return camera_ray_origin(scene.camera)

    def leaf_order(self, scene):
        self.r.upload_scene(scene)
        return [int(o) for o, _ in self.r.bvh_leaf_order()]
