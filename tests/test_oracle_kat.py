"""RNG-free known-answer tests that pin the CPU oracle to the reference's semantics.

The reference ships no tests or golden vectors, and cannot be built here, so every expected value is derived by hand
from the cited reference lines; none depends on a random draw (or, where draws happen, the expected value is the same
for every draw).  The frame-, ray- and host-level cases live in tests/kat_cases.py and run here on the oracle; the SAME
cases run on the HIP path in tests/test_gpu_kat.py (`-m gpu`) and, for the host arithmetic, in tests/test_host_logic.py.
Below them: probes only the oracle offers (AABB::hit alone, the random streams, libm vs polynomial trig)."""
import math

import numpy as np
import pytest

import kat_cases
import pyoracle
import raytracingrust_amd as rt

f32 = np.float32

ORACLE = kat_cases.OracleBackend()


@pytest.mark.parametrize("case", kat_cases.FRAME_CASES + kat_cases.RAY_CASES + kat_cases.HOST_CASES, ids=lambda c: c.__name__)
def test_known_answer_on_the_oracle(case):
    case(ORACLE)


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


def test_rejection_sampler_sub_stream_is_a_uniform_ball_sampler():
    """The rejection sampler's candidates come from a plain 32-bit LCG seeded by one word of the path's stream (DESIGN.md
    section 3) -- this build's definition, shared by kernels and oracle, so GPU-vs-oracle parity says nothing about its
    QUALITY.  What the reference draws is thread_rng's uniform cube (sample.rs:26-51); checked here as a distribution:
    acceptance rate pi/6, octant balance, first and second moments of the accepted points and of the normalised
    directions, and no correlation between successive candidates -- over many streams and inside one long sub-stream."""
    rng_pixels = range(0, 4000, 1)
    pts = np.concatenate([pyoracle.probe_rng(0xBA11, p, p % 7, 3, 96).reshape(-1, 3) for p in rng_pixels])  # 32 calls per stream
    n = len(pts)
    assert n == 4000 * 32
    r2 = (pts.astype(np.float64) ** 2).sum(axis=1)
    assert np.all(r2 <= 1.0 + 1e-6)
    # a uniform ball: E[x] = 0, E[x^2] = 1/5, E[|v|^2] = 3/5, P(|v| <= r) = r^3
    se = 1.0 / math.sqrt(n)
    assert np.all(np.abs(pts.mean(axis=0)) < 5 * math.sqrt(0.2) * se)
    assert np.all(np.abs((pts.astype(np.float64) ** 2).mean(axis=0) - 0.2) < 5 * 0.18 * se)
    assert abs(r2.mean() - 0.6) < 5 * 0.25 * se
    assert abs(np.count_nonzero(r2 <= 0.25) / n - 0.125) < 5 * math.sqrt(0.125 * 0.875) * se
    cov = np.cov(pts.T.astype(np.float64))
    assert np.all(np.abs(cov - np.diag(np.diag(cov))) < 5 * 0.2 * se)
    # octants: chi-square with 7 degrees of freedom (99.99 % quantile 29.9)
    octant = (pts[:, 0] > 0).astype(int) + 2 * (pts[:, 1] > 0) + 4 * (pts[:, 2] > 0)
    counts = np.bincount(octant, minlength=8).astype(np.float64)
    assert ((counts - n / 8) ** 2 / (n / 8)).sum() < 29.9
    # normalised directions (sample_unit_sphere_surface, sample.rs:66-70): E[d] = 0, E[d_i^2] = 1/3
    d = pts.astype(np.float64) / np.sqrt(r2)[:, None]
    assert np.all(np.abs(d.mean(axis=0)) < 5 * math.sqrt(1 / 3) * se)
    assert np.all(np.abs((d ** 2).mean(axis=0) - 1 / 3) < 5 * 0.3 * se)
    # successive calls on one stream are uncorrelated (each call takes ONE word of the path's stream)
    a, b = pts[0::2].astype(np.float64), pts[1::2].astype(np.float64)
    assert np.all(np.abs((a * b).mean(axis=0)) < 5 * 0.2 / math.sqrt(len(a)))

    # inside ONE sub-stream: the candidate coordinates the sampler would test, in order (x0 y0 z0 x1 y1 z1 ...)
    acc, total = 0, 0
    lag = np.zeros(4)
    m = 0
    for p in range(200):
        c = pyoracle.probe_rng(0xBA11, 100000 + p, 0, 2, 3 * 2048).astype(np.float64)
        cand = c.reshape(-1, 3)
        acc += int(np.count_nonzero((cand ** 2).sum(axis=1) <= 1.0))
        total += len(cand)
        for k in range(1, 5):  # serial correlation of the raw coordinate sequence at lags 1..4 (x-y, x-z, x-next x, ...)
            lag[k - 1] += float((c[:-k] * c[k:]).mean())
        m += 1
    rate = acc / total
    assert abs(rate - math.pi / 6) < 5 * math.sqrt((math.pi / 6) * (1 - math.pi / 6) / total), rate
    assert np.all(np.abs(lag / m) < 5 * (1 / 3) / math.sqrt(total * 3)), lag / m
