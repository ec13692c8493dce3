// rtr_kernels.hip -- the gfx950 path-tracing kernels (hand-written HIP for CDNA4, wave64).
//
// One lane = one pixel's path; one 64-lane wave = one 8x8 pixel tile (primary rays of a wave stay
// coherent).  A lane whose path terminates immediately regenerates the next sample of its pixel inside
// the same loop iteration ("lane regeneration"), so every trip of the wave's loop is one ray segment for
// every live lane: path-length divergence never idles lanes, only the material switch and the rejection
// sampler diverge.  Samples of a pixel are accumulated in registers in sample order, so the per-pixel sum
// is the reference's sum (render.rs:84-100) bit for bit.
//
// Arithmetic contract: built with -ffp-contract=off and correctly rounded f32 divide/sqrt, every value
// is computed by the reference's operation sequence (citations per function), so the float framebuffer
// equals the CPU oracle's.  Hoisting is limited to pure functions of the same operands (1/d per ray,
// |d|^2 per ray, r^2 per sphere, triangle edges).
//
// The attenuation product of a path is right-nested in the reference (render.rs:52:
// att0*(att1*(...*terminal))); floats do not reassociate, so each bounce pushes the id of its
// attenuation onto a per-lane chain in LDS and the product is replayed innermost-first on termination.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <type_traits>

#include "rtr_device.h"

namespace rtr {

#define RTR_DEV __device__ __forceinline__

#ifdef RTR_X_LANESTAT // diagnostic build only (scripts/lanestat.py): lanes active per block of the headline kernel's loop
__shared__ uint32_t rtr_lsc[32];
#define RTR_LS(k, cond) { const uint64_t m_ = __ballot(cond); \
        if (m_ && (threadIdx.x & 63u) == (uint32_t)(__ffsll((long long)m_) - 1)) atomicAdd(&rtr_lsc[k], (uint32_t)__popcll(m_)); }
#define RTR_LS1(k) { const uint64_t m_ = __ballot(true); if ((threadIdx.x & 63u) == (uint32_t)(__ffsll((long long)m_) - 1)) atomicAdd(&rtr_lsc[k], 1u); }
#else
#define RTR_LS(k, cond)
#define RTR_LS1(k)
#endif

struct V3 { float x, y, z; };
RTR_DEV V3 mk(float x, float y, float z) { V3 r; r.x = x; r.y = y; r.z = z; return r; }
RTR_DEV V3 ld3(const float *p) { return mk(p[0], p[1], p[2]); }
RTR_DEV V3 add(V3 a, V3 b) { return mk(a.x + b.x, a.y + b.y, a.z + b.z); }
RTR_DEV V3 sub(V3 a, V3 b) { return mk(a.x - b.x, a.y - b.y, a.z - b.z); }
RTR_DEV V3 scale(V3 a, float s) { return mk(a.x * s, a.y * s, a.z * s); }
RTR_DEV V3 divs(V3 a, float s) { return mk(a.x / s, a.y / s, a.z / s); }
RTR_DEV V3 neg(V3 a) { return mk(-a.x, -a.y, -a.z); }
RTR_DEV V3 mul(V3 a, V3 b) { return mk(a.x * b.x, a.y * b.y, a.z * b.z); }
// cgmath 0.18: dot sums left to right
RTR_DEV float dot(V3 a, V3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }
RTR_DEV V3 cross(V3 a, V3 b) {
    return mk((a.y * b.z) - (a.z * b.y), (a.z * b.x) - (a.x * b.z), (a.x * b.y) - (a.y * b.x));
}
// Correctly rounded f32 square root.  hipcc's expansion of sqrtf() is v_sqrt_f32 (1 ulp) + a test of the
// two neighbouring floats with exact fma residuals, wrapped in 2^32 input scaling for denormals and a class
// check for 0/inf/NaN (20 instructions, 10 of them half-rate selects/compares).  For x in [2^-100, 2^100) the
// wrapping is dead weight: this is the same neighbour test without it (9 instructions), sqrtf() otherwise.
// rtr_selftest_sqrt() compares the two over all 2^32 bit patterns on the device (tests/test_gpu_selftest.py).
RTR_DEV float rt_sqrt(float x) {
    const uint32_t b = __float_as_uint(x);
    if ((b - 0x0D800000u) < 0x65000000u) { // sign 0, biased exponent in [27, 228)
        float s = __builtin_amdgcn_sqrtf(x);
        const float lo = __uint_as_float(__float_as_uint(s) - 1u), hi = __uint_as_float(__float_as_uint(s) + 1u);
        const float rl = __builtin_fmaf(-lo, s, x), rh = __builtin_fmaf(-hi, s, x);
        if (rl <= 0.0f) s = lo;
        if (rh > 0.0f) s = hi;
        return s;
    }
    return sqrtf(x);
}

// The fast half of rt_sqrt alone, for straight-line code: the correctly rounded root when `in_range` comes back true (x in
// [2^-100, 2^100)), anything otherwise -- the caller has an exact path of its own for those.
RTR_DEV float rt_sqrt_core(float x, bool &in_range) {
    const uint32_t b = __float_as_uint(x);
    in_range = (b - 0x0D800000u) < 0x65000000u;
    float s = __builtin_amdgcn_sqrtf(x);
    const float lo = __uint_as_float(__float_as_uint(s) - 1u), hi = __uint_as_float(__float_as_uint(s) + 1u);
    const float rl = __builtin_fmaf(-lo, s, x), rh = __builtin_fmaf(-hi, s, x);
    s = rl <= 0.0f ? lo : s;
    s = rh > 0.0f ? hi : s;
    return s;
}

// cgmath normalize = v * (1/|v|)
RTR_DEV V3 normalize(V3 a) { return scale(a, 1.0f / rt_sqrt(dot(a, a))); }

constexpr float kPi = 3.14159265358979323846f;
constexpr float kTMin = 0.00001f; // render.rs:38
constexpr uint32_t kChainClay = 0xFFFFFFFFu;

// --- counter-based random stream (DESIGN.md "Random numbers") ---------------------------------------
struct Rng { uint32_t state, inc; };
RTR_DEV uint32_t hash32(uint32_t x) {
    x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16;
    return x;
}
RTR_DEV uint32_t rng_u32(Rng &r) {
    uint32_t old = r.state;
    r.state = old * 747796405u + r.inc;
    uint32_t word = ((old >> ((old >> 28) + 4u)) ^ old) * 277803737u;
    return (word >> 22) ^ word;
}
// rand 0.8 float conversions (top 23 bits k of a u32).  The reference forms k*2^-23 with an int->float
// convert and scales it; here the same VALUES are assembled in the mantissa: (1.m) - 1 = k*2^-23 and
// (2 + k*2^-22) - 3 = k*2^-22 - 1 are exact in f32, so the results are bit-identical to
// value0_1*scale + low (the oracle computes them that way) at half the instructions.
RTR_DEV float rng_value01(Rng &r) { return __uint_as_float(0x3f800000u | (rng_u32(r) >> 9)) - 1.0f; }
RTR_DEV float rng_unit_inclusive(Rng &r) { return rng_value01(r) * 0x1.000002p+0f; } // Uniform::from(0.0..=1.0), render.rs:81
RTR_DEV float rng_unit(Rng &r) { return rng_value01(r); }                            // gen_range(0.0..1.0), materials.rs:127,151

// sample.rs:43-51 rejection-sampled unit ball; returns |v|^2 through m2.  The candidates of ONE call come from a
// sub-stream of their own: one word r of the path's stream seeds a plain 32-bit LCG (the PCG multiplier, a fixed odd
// increment) whose successive states give x, y, z of candidate 0, 1, 2, ... through their top 23 bits -- rand 0.8's
// gen_range(-1.0..1.0) grid, as before.  The path's stream therefore advances by exactly one word per call however
// many candidates are rejected, candidate k is a closed-form function of (r, k) (LCG jump-ahead), and a candidate
// costs three multiply-adds instead of three PCG output permutations: the rejection loop is the most divergent block
// of the headline kernel (11 of 64 lanes active per wave iteration, scripts/lanestat.py).
constexpr uint32_t kBallMul = 747796405u, kBallInc = 2891336453u;
RTR_DEV float ball_coord(uint32_t &s) {
    s = s * kBallMul + kBallInc;
    return __uint_as_float(__builtin_amdgcn_alignbit(0x80u, s, 9u)) - 3.0f; // 0x40000000 | (s >> 9): [2, 4) - 3
}
RTR_DEV V3 sample_unit_ball(Rng &rng, float &m2) {
    uint32_t s = rng_u32(rng);
    V3 v;
    do {
        v.x = ball_coord(s);
        v.y = ball_coord(s);
        v.z = ball_coord(s);
        m2 = dot(v, v);
    } while (!(m2 <= 1.0f));
    return v;
}
// sample.rs:66-70
RTR_DEV V3 sample_unit_sphere(Rng &rng) {
    float m2;
    V3 v = sample_unit_ball(rng, m2);
    return scale(v, 1.0f / rt_sqrt(m2));
}

// --- polynomial acos / atan2 for SkyMap (same sequence as the oracle's OR_TRIG_POLY) -----------------
RTR_DEV float poly_asin_core(float x) {
    float z = x * x;
    float p = 4.2163199048E-2f;
    p = p * z + 2.4181311049E-2f;
    p = p * z + 4.5470025998E-2f;
    p = p * z + 7.4953002686E-2f;
    p = p * z + 1.6666752422E-1f;
    return (p * z) * x + x;
}
RTR_DEV float poly_acos(float x) {
    float a = fabsf(x);
    if (a <= 0.5f) return 1.57079632679489661923f - poly_asin_core(x);
    float z = (1.0f - a) * 0.5f;
    float r = 2.0f * poly_asin_core(rt_sqrt(z)); // (rt_sqrt == sqrtf bit for bit: rtr_selftest_sqrt)
    return x > 0.0f ? r : kPi - r;
}
RTR_DEV float poly_atan_pos(float x) {
    float y0 = 0.0f;
    if (x > 2.414213562373095f) { y0 = 1.57079632679489661923f; x = -(1.0f / x); }
    else if (x > 0.4142135623730950f) { y0 = 0.78539816339744830962f; x = (x - 1.0f) / (x + 1.0f); }
    float z = x * x;
    float p = 8.05374449538e-2f;
    p = p * z - 1.38776856032E-1f;
    p = p * z + 1.99777106478E-1f;
    p = p * z - 3.33329491539E-1f;
    return y0 + ((p * z) * x + x);
}
RTR_DEV float poly_atan2(float y, float x) {
    if (!(x == x) || !(y == y)) return __builtin_nanf("");
    if (y == 0.0f) return (x < 0.0f || (x == 0.0f && __builtin_signbit(x))) ? __builtin_copysignf(kPi, y) : y;
    if (x == 0.0f) return y > 0.0f ? 1.57079632679489661923f : -1.57079632679489661923f;
    float r = poly_atan_pos(fabsf(y) / fabsf(x));
    if (x < 0.0f) r = kPi - r;
    return y < 0.0f ? -r : r;
}

// Rust `as usize` (backgrounds.rs:88-89): saturating, NaN -> 0
RTR_DEV unsigned long long sat_usize(float f) {
    if (!(f == f)) return 0ull;
    if (f <= 0.0f) return 0ull;
    if (f >= 18446744073709551616.0f) return ~0ull;
    return (unsigned long long)f;
}
// Rust `as u8` (color.rs:34-41)
RTR_DEV uint32_t sat_u8(float f) {
    if (!(f == f)) return 0u;
    if (f <= 0.0f) return 0u;
    if (f >= 255.0f) return 255u;
    return (uint32_t)f;
}

// Division by a denominator that many numerators share.  hipcc's correctly rounded n / d is
//     d' = div_scale(d); r0 = rcp(d'); e0 = fma(-d', r0, 1); r1 = fma(e0, r0, r0);            -- depends on d only
//     n' = div_scale(n); q0 = n' * r1; e1 = fma(-d', q0, n'); q1 = fma(e1, r1, q0); e2 = fma(-d', q1, n');
//     q = div_fmas(e2, r1, q1); div_fixup(q, d, n)
// where div_scale / div_fmas only rescale operands with extreme exponents and div_fixup only patches zero, infinite
// and NaN operands.  For 2^-20 <= |d| <= 2^40 and n == +0 or 2^-40 <= |n| <= 2^60 none of them changes a bit, so the
// quotient is the five FMA-class operations below on a reciprocal computed once: the same operations in the same
// order, hence the same correctly rounded result (rtr_selftest_div sweeps numerator ranges against `/` on the device).
struct SharedDiv {
    float d, r1;
    bool ok; // the denominator is in the range above
};
RTR_DEV SharedDiv shared_div_prepare(float d) {
    SharedDiv s;
    s.d = d;
    const float r0 = __builtin_amdgcn_rcpf(d);
    const float e0 = __builtin_fmaf(-d, r0, 1.0f);
    s.r1 = __builtin_fmaf(e0, r0, r0);
    s.ok = fabsf(d) >= 0x1p-20f && fabsf(d) <= 0x1p40f;
    return s;
}
RTR_DEV float shared_div(float n, float d, float r1) {
    const float q0 = n * r1;
    const float e1 = __builtin_fmaf(-d, q0, n);
    const float q1 = __builtin_fmaf(e1, r1, q0);
    const float e2 = __builtin_fmaf(-d, q1, n);
    return __builtin_fmaf(e2, r1, q1);
}
RTR_DEV bool shared_div_numerators_ok(V3 n) { // every component in [2^-40, 2^60] (a zero takes the slow path: rare)
    const float lo = fminf(fminf(fabsf(n.x), fabsf(n.y)), fabsf(n.z)), hi = fmaxf(fmaxf(fabsf(n.x), fabsf(n.y)), fabsf(n.z));
    return lo >= 0x1p-40f && hi <= 0x1p60f;
}

// --- intersection ------------------------------------------------------------------------------------
struct Ray { V3 o, d; };

// aabb.rs:81-137 with 1/d hoisted per ray.  Returns hit; *entry = the narrowed tmin.
// The reference narrows [tmin, tmax] per axis with `if t0 > tmin { tmin = t0 }`, `if t1 < tmax { tmax = t1 }` and
// returns false as soon as `tmax <= tmin`.  Restated without changing any outcome:
//   * the running tmin / tmax are never NaN (tmin starts at 1e-5, tmax at +inf or a hit distance; a NaN t0 / t1
//     -- 0 * inf when the direction component is 0 and the origin lies on the slab plane -- fails its comparison
//     and is ignored), so the conditional updates are exactly fmaxf / fminf (IEEE maxNum / minNum drop the NaN);
//   * tmin only grows and tmax only shrinks, so "tmax <= tmin after some axis" is equivalent to
//     "tmax <= tmin after the last axis": one comparison instead of three.
RTR_DEV bool slab(const float *bmin, const float *bmax, const Ray &r, V3 inv, float tmax, float &entry) {
    float tmin = kTMin;
    float t0 = (bmin[0] - r.o.x) * inv.x, t1 = (bmax[0] - r.o.x) * inv.x;
    tmin = fmaxf(tmin, inv.x < 0.0f ? t1 : t0); // `if inv_d < 0 { swap(t0, t1) }`
    tmax = fminf(tmax, inv.x < 0.0f ? t0 : t1);
    t0 = (bmin[1] - r.o.y) * inv.y; t1 = (bmax[1] - r.o.y) * inv.y;
    tmin = fmaxf(tmin, inv.y < 0.0f ? t1 : t0);
    tmax = fminf(tmax, inv.y < 0.0f ? t0 : t1);
    t0 = (bmin[2] - r.o.z) * inv.z; t1 = (bmax[2] - r.o.z) * inv.z;
    tmin = fmaxf(tmin, inv.z < 0.0f ? t1 : t0);
    tmax = fminf(tmax, inv.z < 0.0f ? t0 : t1);
    entry = tmin;
    return !(tmax <= tmin);
}

// objects.rs:22-51 root selection.  a = |d|^2.  Returns true with t when a root lies in [tmin, tmax].
RTR_DEV bool sphere_range(V3 c, float r2, const Ray &r, float a, float tmin, float tmax, float &t) {
    V3 oc = sub(r.o, c);
    float half_b = dot(oc, r.d);
    float cc = dot(oc, oc) - r2;
    float disc = half_b * half_b - a * cc;
    if (disc < 0.0f) return false;
    float sq = rt_sqrt(disc);
    float root = (-half_b - sq) / a;
    // objects.rs:39-44: `if root < tmin || tmax < root { second root ... }`.  When the FIRST root is already
    // beyond tmax the second (>= the first: sq >= 0, a >= 0) is too, so its division is skipped; a NaN root
    // fails both comparisons and is accepted, exactly as in the reference.
    if (tmax < root) return false;
    if (root < tmin) {
        root = (-half_b + sq) / a;
        if (root < tmin || tmax < root) return false;
    }
    t = root;
    return true;
}
// Sphere::hit as trace_ray reaches it: tmin = 0.00001 (render.rs:38)
RTR_DEV bool sphere_t(V3 c, float r2, const Ray &r, float a, float tmax, float &t) {
    return sphere_range(c, r2, r, a, kTMin, tmax, t);
}

// cephes-style logf (restated identically in the oracle's OR_TRIG_POLY mode) for Volume::hit's ln (volume.rs:58)
RTR_DEV float poly_log(float xx) {
    if (!(xx > 0.0f)) return xx == 0.0f ? -__builtin_inff() : __builtin_nanf("");
    if (xx == __builtin_inff()) return xx;
    uint32_t b = __float_as_uint(xx);
    int e = (int)((b >> 23) & 0xffu) - 126;
    if (((b >> 23) & 0xffu) == 0u) return logf(xx); // denormal: never produced by the 23-bit uniform
    float x = __uint_as_float((b & 0x007fffffu) | 0x3f000000u); // frexp mantissa in [0.5, 1)
    if (x < 0.707106781186547524f) { e -= 1; x = x + x - 1.0f; } else { x = x - 1.0f; }
    float z = x * x;
    float y = 7.0376836292E-2f;
    y = y * x - 1.1514610310E-1f;
    y = y * x + 1.1676998740E-1f;
    y = y * x - 1.2420140846E-1f;
    y = y * x + 1.4249322787E-1f;
    y = y * x - 1.6668057665E-1f;
    y = y * x + 2.0000714765E-1f;
    y = y * x - 2.4999993993E-1f;
    y = y * x + 3.3333331174E-1f;
    y = (y * x) * z;
    const float fe = (float)e;
    if (e) y += -2.12194440e-4f * fe;
    y += -0.5f * z;
    z = x + y;
    if (e) z += 0.693359375f * fe;
    return z;
}

// Volume::hit(ray, tmin, tmax) (lib/volume.rs:34-73) after its two boundary hits t1, t2 (volume.rs:35-36).  Draws one random
// number per call -- once the clipped interval is known to be non-empty -- exactly where the reference draws it.
RTR_DEV bool volume_finish(float t1, float t2, float neg_inv_density, float a, float tmin, float tmax, Rng &rng, float &t) {
    if (t1 < tmin) t1 = tmin;
    if (t2 > tmax) t2 = tmax;
    if (t1 >= t2) return false;
    if (t1 < 0.0f) t1 = 0.0f;
    const float ray_length = rt_sqrt(a); // ray.direction.magnitude()
    const float distance_inside_boundary = (t2 - t1) * ray_length;
    const float random_float = rng_unit_inclusive(rng); // gen_range(0.0..=1.0), volume.rs:57
    const float hit_distance = neg_inv_density * poly_log(random_float);
    if (hit_distance > distance_inside_boundary) return false;
    t = t1 + hit_distance / ray_length;
    return true;
}
// ... over a sphere boundary
RTR_DEV bool volume_t(V3 c, float r2, float neg_inv_density, const Ray &r, float a, float tmin, float tmax, Rng &rng, float &t) {
    float t1, t2;
    if (!sphere_range(c, r2, r, a, -__builtin_inff(), __builtin_inff(), t1)) return false;
    if (!sphere_range(c, r2, r, a, t1 + tmin, __builtin_inff(), t2)) return false;
    return volume_finish(t1, t2, neg_inv_density, a, tmin, tmax, rng, t);
}
// mesh.rs:146-184 Moeller-Trumbore with the reference's tests (tmax ignored).
RTR_DEV bool triangle_t(V3 v0, V3 e1, V3 e2, const Ray &r, float &t, float &u, float &v) {
    V3 h = cross(r.d, e2);
    float a = dot(e1, h);
    if (a > -kTMin && a < kTMin) return false;
    float f = 1.0f / a;
    V3 s = sub(r.o, v0);
    float uu = f * dot(s, h);
    if (uu < 0.0f || uu > 1.0f) return false;
    V3 q = cross(s, e1);
    float vv = f * dot(r.d, q);
    if (vv < 0.0f || uu + vv > 1.0f) return false;
    float tt = f * dot(e2, q);
    if (!(tt > kTMin)) return false;
    t = tt; u = uu; v = vv;
    return true;
}

// --- Volume over a mesh boundary (lib/volume.rs:20 `boundary: Box<dyn Object>`) ----------------------------------------
// Volume::hit calls boundary.hit twice (volume.rs:35-36), with [-inf, +inf] and with [t1 + tmin, +inf].  For a Mesh that
// is bvh_root.hit(ray, tmin, tmax) (mesh.rs:124-127): every node's box is tested with the caller's range (AABB::hit,
// aabb.rs:81-137) and Triangle::hit takes tmin both as its determinant epsilon and as the lower bound on t, ignoring tmax
// (mesh.rs:166,190,151).  The ranges start below zero, so the walk's `kTMin` forms and its culling do not apply: these are
// the general forms, and the nested walk below visits every pierced box, as the reference does.
RTR_DEV bool slab_range(const float *bmin, const float *bmax, const Ray &r, V3 inv, float tmin, float tmax) {
    float t0 = (bmin[0] - r.o.x) * inv.x, t1 = (bmax[0] - r.o.x) * inv.x;
    tmin = fmaxf(tmin, inv.x < 0.0f ? t1 : t0);
    tmax = fminf(tmax, inv.x < 0.0f ? t0 : t1);
    t0 = (bmin[1] - r.o.y) * inv.y; t1 = (bmax[1] - r.o.y) * inv.y;
    tmin = fmaxf(tmin, inv.y < 0.0f ? t1 : t0);
    tmax = fminf(tmax, inv.y < 0.0f ? t0 : t1);
    t0 = (bmin[2] - r.o.z) * inv.z; t1 = (bmax[2] - r.o.z) * inv.z;
    tmin = fmaxf(tmin, inv.z < 0.0f ? t1 : t0);
    tmax = fminf(tmax, inv.z < 0.0f ? t0 : t1);
    return !(tmax <= tmin);
}
RTR_DEV bool triangle_range(V3 v0, V3 e1, V3 e2, const Ray &r, float tmin, float &t) {
    V3 h = cross(r.d, e2);
    float a = dot(e1, h);
    if (a > -tmin && a < tmin) return false;
    float f = 1.0f / a;
    V3 s = sub(r.o, v0);
    float uu = f * dot(s, h);
    if (uu < 0.0f || uu > 1.0f) return false;
    V3 q = cross(s, e1);
    float vv = f * dot(r.d, q);
    if (vv < 0.0f || uu + vv > 1.0f) return false;
    float tt = f * dot(e2, q);
    if (!(tt > tmin)) return false;
    t = tt;
    return true;
}
// Mesh::hit(ray, tmin, tmax) over the flattened subtree `root` (its own box in bmin / bmax).  The lane's traversal stack is
// used above `sp0` (the outer walk's entries stay untouched); refs only -- nothing is culled, so no entry distances.
RTR_DEV bool mesh_range_hit(const DevNode *nodes, const DevLeafHot *leaf_hot, uint32_t root, V3 bmin_v, V3 bmax_v, const Ray &r, V3 inv,
                            float tmin, float tmax, uint32_t *stk_ref, uint32_t sp0, uint32_t lane, float &t_out) {
    const float bmin[3] = {bmin_v.x, bmin_v.y, bmin_v.z}, bmax[3] = {bmax_v.x, bmax_v.y, bmax_v.z};
    if (root == REF_NONE) return false;
    if ((root & (REF_LEAF | REF_PAIR)) != REF_LEAF && !slab_range(bmin, bmax, r, inv, tmin, tmax)) return false;
    bool found = false;
    float bt = 0.0f;
    uint32_t bl = 0u;
    uint32_t sp = sp0, cur = root;
    for (;;) {
        bool pop = false;
        if (cur & REF_LEAF) {
            const uint32_t L = cur & REF_INDEX, n = (cur & REF_PAIR) ? 2u : 1u;
            for (uint32_t k = 0; k < n; k++) {
                const float4 *hot = reinterpret_cast<const float4 *>(leaf_hot + (L + k));
                const float4 q0 = hot[0], q1 = hot[1], q2 = hot[2];
                float t;
                // merge_optionals (bvh.rs:181-186): the smaller t wins, the later leaf in depth-first order wins exact ties
                if (triangle_range(mk(q0.x, q0.y, q0.z), mk(q0.w, q1.x, q1.y), mk(q1.z, q1.w, q2.x), r, tmin, t) &&
                    (!found || t < bt || (t == bt && L + k > bl))) {
                    found = true; bt = t; bl = L + k;
                }
            }
            pop = true;
        } else {
            const float4 *nq = reinterpret_cast<const float4 *>(nodes + cur);
            const float4 q0 = nq[0], q1 = nq[1], q2 = nq[2], q3 = nq[3];
            const float lmin[3] = {q0.x, q0.y, q0.z}, lmax[3] = {q0.w, q1.x, q1.y};
            const float rmin[3] = {q1.z, q1.w, q2.x}, rmax[3] = {q2.y, q2.z, q2.w};
            const uint32_t l = __float_as_uint(q3.x), rr = __float_as_uint(q3.y);
            bool hl = l != REF_NONE, hr = rr != REF_NONE;
            if (hl && (l & (REF_LEAF | REF_PAIR)) != REF_LEAF) hl = slab_range(lmin, lmax, r, inv, tmin, tmax);
            if (hr && (rr & (REF_LEAF | REF_PAIR)) != REF_LEAF) hr = slab_range(rmin, rmax, r, inv, tmin, tmax);
            if (hl && hr) {
                stk_ref[sp * 64 + lane] = rr;
                sp++;
                cur = l;
            } else if (hl) {
                cur = l;
            } else if (hr) {
                cur = rr;
            } else {
                pop = true;
            }
        }
        if (pop) {
            if (sp == sp0) break;
            sp--;
            cur = stk_ref[sp * 64 + lane];
        }
    }
    t_out = bt;
    return found;
}

// hipcc sinks the individual 16-B loads of a node / leaf record into the branches that consume them, which turns
// one gather into two or three DEPENDENT round trips on a latency-bound loop.  Naming every component as an input
// of one empty asm statement pins all loads of the record in front of it: they issue together, one wait covers them.
RTR_DEV void gather_fence(const float4 &a, const float4 &b, const float4 &c) {
    asm volatile("" ::"v"(a.x), "v"(a.y), "v"(a.z), "v"(a.w), "v"(b.x), "v"(b.y), "v"(b.z), "v"(b.w), "v"(c.x), "v"(c.y), "v"(c.z), "v"(c.w));
}
RTR_DEV void gather_fence(const float4 &a, const float4 &b, const float4 &c, const float2 &d) {
    asm volatile("" ::"v"(a.x), "v"(a.y), "v"(a.z), "v"(a.w), "v"(b.x), "v"(b.y), "v"(b.z), "v"(b.w), "v"(c.x), "v"(c.y), "v"(c.z), "v"(c.w),
                 "v"(d.x), "v"(d.y));
}
RTR_DEV void gather_fence(const float4 &a, const float4 &b, float c) {
    asm volatile("" ::"v"(a.x), "v"(a.y), "v"(a.z), "v"(a.w), "v"(b.x), "v"(b.y), "v"(b.z), "v"(b.w), "v"(c));
}
RTR_DEV void gather_fence(const float4 &a, const float4 &b, const float4 &c, const float4 &d) {
    asm volatile("" ::"v"(a.x), "v"(a.y), "v"(a.z), "v"(a.w), "v"(b.x), "v"(b.y), "v"(b.z), "v"(b.w), "v"(c.x), "v"(c.y), "v"(c.z), "v"(c.w),
                 "v"(d.x), "v"(d.y), "v"(d.z), "v"(d.w));
}

// Entry distances on the traversal stack.  They are read back for one purpose: to skip a deferred box that the best hit
// has overtaken meanwhile (`e > best.cull`).  Any value not above the true entry distance keeps that test exact -- it can
// only fail to skip, and the box's children are then tested against the best hit anyway -- so the stack keeps the upper
// 16 bits of the f32 (truncation: toward zero; entries are positive, or +0 for "never skip").  Six bytes per
// entry instead of eight: LDS per wave is what limits the tree kernels' occupancy (house.obj: 17 -> 23 waves per CU).
// Small trees (at most 65 536 node records and 65 536 leaves: every BASELINE scene) pack an entry into ONE word: the ref's
// four flag bits, a 16-bit index and the upper 12 bits of the distance (sign dropped: entries are never negative).  Four
// bytes per entry instead of six, one LDS instruction per push and pop instead of two.  LDS is granted in 1280-byte
// steps on gfx950 (measured: DESIGN.md section 5.2), so house.obj's 14 entries + tables went from 5632 B = 5 steps
// = 25 waves per CU to 3840 B = 3 steps, where the 72 VGPRs' 28 waves fit.
RTR_DEV uint32_t stack_pack32(uint32_t ref, float e) {
    return (ref & 0xF0000000u) | ((ref & 0xFFFFu) << 12) | ((__float_as_uint(e) >> 19) & 0xFFFu);
}
RTR_DEV uint32_t stack_ref32(uint32_t w) { return (w & 0xF0000000u) | ((w >> 12) & 0xFFFFu); }
RTR_DEV float stack_t32(uint32_t w) { return __uint_as_float((w & 0xFFFu) << 19); }
typedef uint16_t StackT;
RTR_DEV StackT stack_t_pack(float e) { return (StackT)(__float_as_uint(e) >> 16); }
RTR_DEV float stack_t_unpack(StackT h) { return __uint_as_float((uint32_t)h << 16); }

// The walk's best hit so far.  Two registers: the barycentrics of a winning triangle are re-derived once, in finish_leaf,
// from the same operands by the same operations (round 2 carried u, v and the cull bound through every trip; the tree
// kernels live at the edge of their register budget -- DESIGN.md section 5.2).
struct Best {
    float t;       // +inf until a leaf is hit
    uint32_t leaf; // REF_LEAF|type|L, or REF_NONE
};
// The reference visits every box the ray pierces (bvh.rs:160-166); the walk here skips a box whose entry distance lies
// beyond the best hit so far.  Entry distances and hit distances are rounded results of different formulas, so a
// primitive's computed t can undercut its own box's computed entry by a few ulps -- most visibly where a sphere touches
// its box, or through the cancellation in a large sphere's root (r = 100: ~1e-5 absolute).  A box is therefore skipped
// only when it is entered beyond the best hit by a relative margin of 2^-10, far above those errors and far below any
// difference that would save work: with it every tree frame of the test-suite, at every size, is bit-identical to the
// oracle's unculled recursive traversal (tests/test_gpu_parity.py); without it 0-20 pixels per 1080p frame differed in
// the last bit.  Cost of the margin: below measurement noise.
constexpr float kCullMargin = 1.0009765625f; // 1 + 2^-10
RTR_DEV void best_reset(Best &b) { b.leaf = REF_NONE; b.t = __builtin_inff(); }
RTR_DEV void best_take(Best &b, float t, uint32_t leaf) { b.t = t; b.leaf = leaf; }
// boxes entered beyond this distance are skipped: +inf until a leaf is hit; t > 0 (every accepted t exceeds tmin); a NaN t
// gives a NaN bound: nothing is culled
RTR_DEV float best_cull(const Best &b) { return b.t * kCullMargin; }

// Traversal counters of a lane (rtr_stats node_steps / leaf_tests / sky_lookups, option "traversal_counters"): incremented
// where the walk does the work; dead code -- and removed by the compiler -- in every kernel that does not store them.
struct WalkCounts { uint32_t nodes = 0u, leaves = 0u, sky = 0u; };

// Candidate acceptance inside one tree: the reference merges with `left < right ? left : right`
// (bvh.rs:181-186): the smaller t wins, and the right operand -- the later leaf in depth-first order -- wins whenever
// the left one is not strictly smaller: exact ties, and any comparison with a NaN distance (a sphere hit by a ray with
// non-finite components: `disc = NaN` slips through objects.rs:31).  Pairwise this is exactly the merge; over a whole
// tree it equals the reference's nested merges when the hits are all finite or all NaN (a NaN ray makes every sphere's
// distance NaN, and the last hit leaf wins as in the reference); a MIX of NaN and finite distances on one ray is not
// an order, the nested merges then depend on the tree's shape, and that case alone stays a documented deviation.
RTR_DEV bool accept_tree(const Best &b, float t, uint32_t leaf) {
    return b.leaf == REF_NONE || t < b.t || (!(b.t < t) && (leaf & REF_INDEX) > (b.leaf & REF_INDEX));
}

// Volume::hit(ray, tmin, tmax) for the Volume leaf record q0..q2 (rtr_device.h) with density word `nid`: sphere boundary, or
// a mesh boundary walked on the lane's stack above `sp`.
template <uint32_t VOL>
RTR_DEV bool volume_boundary_t(const RenderParams &p, const float4 &q0, const float4 &q1, const float4 &q2, float nid, const Ray &r, V3 inv, float a,
                               float tmin, float tmax, Rng &rng, uint32_t *stk_ref, uint32_t sp, uint32_t lane, float &t) {
    // VOL == 1: every Volume of the scene has a sphere boundary, and the nested walks are not compiled in (they cost the
    // launch-bounded tree kernels their register budget); VOL == 2: the leaf record says which
    if (VOL < 2u || (__float_as_uint(q1.z) & 0xFFu) == 0u) return volume_t(mk(q0.x, q0.y, q0.z), q1.x, nid, r, a, tmin, tmax, rng, t);
    const V3 bmin = mk(q0.x, q0.y, q0.z), bmax = mk(q2.y, q2.z, q2.w);
    const uint32_t root = __float_as_uint(q1.w);
    float t1, t2;
    if (!mesh_range_hit(p.nodes, p.leaf_hot, root, bmin, bmax, r, inv, -__builtin_inff(), __builtin_inff(), stk_ref, sp, lane, t1)) return false;
    if (!mesh_range_hit(p.nodes, p.leaf_hot, root, bmin, bmax, r, inv, t1 + tmin, __builtin_inff(), stk_ref, sp, lane, t2)) return false;
    return volume_finish(t1, t2, nid, a, tmin, tmax, rng, t);
}
// ... as world.hit reaches it (tmin = 0.00001).  A Volume whose boundary is a Volume (volume.rs:20: any Object; bit 8 of the
// record's kind word, the inner density in q2.x): the two boundary.hit calls of volume.rs:35-36 are Volume::hit again, with
// the ranges [-inf, +inf] and [t1 + tmin, +inf] -- each draws its own random number, in call order, before the outer one.
template <uint32_t VOL>
RTR_DEV bool volume_leaf_t(const RenderParams &p, const float4 &q0, const float4 &q1, const float4 &q2, const Ray &r, V3 inv, float a,
                           float tmax, Rng &rng, uint32_t *stk_ref, uint32_t sp, uint32_t lane, float &t) {
    if (VOL < 2u || (__float_as_uint(q1.z) & 0x100u) == 0u)
        return volume_boundary_t<VOL>(p, q0, q1, q2, q1.y, r, inv, a, kTMin, tmax, rng, stk_ref, sp, lane, t);
    float t1, t2;
    if (!volume_boundary_t<VOL>(p, q0, q1, q2, q2.x, r, inv, a, -__builtin_inff(), __builtin_inff(), rng, stk_ref, sp, lane, t1)) return false;
    if (!volume_boundary_t<VOL>(p, q0, q1, q2, q2.x, r, inv, a, t1 + kTMin, __builtin_inff(), rng, stk_ref, sp, lane, t2)) return false;
    return volume_finish(t1, t2, q1.y, a, kTMin, tmax, rng, t);
}

template <uint32_t VOL>
RTR_DEV void test_leaf(const RenderParams &p, uint32_t ref, const Ray &r, V3 inv, float a, Best &best, Rng &rng, uint32_t *stk_ref,
                       uint32_t sp, uint32_t lane) {
    (void)inv; (void)stk_ref; (void)sp; (void)lane;
    // A wave's gather of per-lane records costs the vector L1 ~43 clocks for every 16-byte load instruction and next to
    // nothing for an 8- or 4-byte one (scripts/ubench/gather.hip: 48 B as 3 x 16 -> 71 ns per wave-gather and CU, 36 B as
    // 16 + 16 + 4 -> 54 ns), and the tree kernels are bound by exactly that: each kind of leaf loads what it reads.
    const float4 *hot = reinterpret_cast<const float4 *>(p.leaf_hot + (ref & REF_INDEX));
    if (ref & REF_TRI) {
        const float4 q0 = hot[0], q1 = hot[1];
        const float q2x = reinterpret_cast<const float *>(hot)[8];
        gather_fence(q0, q1, q2x);
        float t, u, v;
        // (Möller-Trumbore without its early exits -- all four tests on every lane -- was measured neutral.)
        if (triangle_t(mk(q0.x, q0.y, q0.z), mk(q0.w, q1.x, q1.y), mk(q1.z, q1.w, q2x), r, t, u, v) &&
            accept_tree(best, t, ref)) {
            best_take(best, t, ref);
        }
    } else if (VOL && (ref & REF_VOL)) {
        const float4 q0 = hot[0], q1 = hot[1], q2 = hot[2];
        gather_fence(q0, q1, q2);
        float t;
        // Volume::hit(ray, tmin, +inf) as called from the tree (bvh.rs:156)
        if (volume_leaf_t<VOL>(p, q0, q1, q2, r, inv, a, __builtin_inff(), rng, stk_ref, sp, lane, t) && accept_tree(best, t, ref)) {
            best_take(best, t, ref);
        }
    } else {
        const float4 q0 = hot[0]; // centre, radius^2: one load (the radius itself is the record's fifth word, read by finish_leaf)
        float t;
        // Sphere::hit(ray, tmin, +inf) as called from the tree (bvh.rs:156)
        if (sphere_t(mk(q0.x, q0.y, q0.z), q0.w, r, a, __builtin_inff(), t) && accept_tree(best, t, ref)) {
            best_take(best, t, ref);
        }
    }
}

// A per-lane condition as the wave's 64-bit lane mask (a scalar register pair: what the compare wrote) and back.
__device__ __forceinline__ uint64_t wave_mask(bool c) { return __builtin_amdgcn_ballot_w64(c); }
__device__ __forceinline__ bool lane_of(uint64_t mask) { return __builtin_amdgcn_inverse_ballot_w64(mask); }


// One trip of BvhNode::hit (bvh.rs:146-172) over the flattened tree for a lane whose walk is in progress: a node step (both
// children's boxes from one 64-byte record) unless the lane stands at a leaf, then a leaf test if it stands at one now (the
// second leaf of a collapsed pair follows in the next trip, no stack traffic), then, when the lane has nowhere to descend,
// a pop.  The reference visits both children of
// every pierced box with the caller's [tmin, tmax]; here children are visited nearest-entry first and a box is skipped
// when it is entered beyond the best hit (Best::cull) -- nothing inside it can win or tie.
// Volume leaves draw a random number on every visit (volume.rs:57), so a subtree that holds one is visited exactly when,
// and in the order, the reference visits it: never culled by the best hit, left before right.
// The body exists ONCE, as a macro: the wave-synchronous walk (`traverse`), the resumable walk (k_render_bvh), the
// sample-parallel walk (k_render_bvh_sp) and the streaming walk (k_wf_intersect) expand it in place -- as a function
// shared by several callers it compiled 6 ... 18 % slower in each of them.  Names it expects in scope: p, lane, cur, sp,
// best, inv, a, rng, stk_ref, stk_t, the template flags VOL, MESH and stack_packed (constexpr bool: one-word stack entries).
//   RAY       the ray being walked
//   TMAX_BOX  the tmax the reference hands to AABB::hit: +inf from the scene tree, closest_so_far when a mesh is reached
//             from the linear world (scene.rs:147, mesh.rs:124-127)
//   ON_EMPTY  statement executed when the stack runs empty: the walk is over
// (Measured and not adopted: one kind of step -- node or leaf -- per trip, 0 ... -10 %; a while-while form, -20 %; both
// triangles of a pair in one 96-byte gather, +13 VGPRs and no faster; an explicit bound test on the push, -1 ... -4 %:
// one push per node record on the path, sp <= tree height < p.stack_cap by construction of the flattener, checked on the
// host by tests/cpp/scene_sanity.cpp.)
// The node step of a trip: both children's boxes (q0_ .. q2_ and the refs / flags in q3_: one DevNode record) against the
// ray, the nearer child taken, the other deferred on the lane's stack; sets pop_ when neither is entered.
#define RTR_NODE_STEP(RAY, TMAX_BOX, q0_, q1_, q2_, q3_)                                                                   \
    {                                                                                                                    \
        counts.nodes++;                                                                                                  \
        float lmin_[3] = {q0_.x, q0_.y, q0_.z}, lmax_[3] = {q0_.w, q1_.x, q1_.y};                                    \
        float rmin_[3] = {q1_.z, q1_.w, q2_.x}, rmax_[3] = {q2_.y, q2_.z, q2_.w};                                    \
        const uint32_t l_ = __float_as_uint(q3_.x), r_ = __float_as_uint(q3_.y);                                     \
        const uint32_t flags_ = __float_as_uint(q3_.z);                                                              \
        const bool lvol_ = VOL && (flags_ & NODE_LVOL), rvol_ = VOL && (flags_ & NODE_RVOL);                         \
        bool hl_ = l_ != REF_NONE, hr_ = r_ != REF_NONE;                                                             \
        float el_ = 0.0f, er_ = 0.0f;                                                                                \
        if constexpr (MESH) {                                                                                        \
            /* Mesh scenes: nearly every child is a Split node or a leaf pair, i.e. carries a box.  Both slab tests */ \
            /* run unconditionally, side by side -- straight-line code instead of two exec-mask regions            */ \
            /* (+2 ... +8 % on C3/C4; on sphere scenes, where half the children are bare leaves, it loses 4 %).    */ \
            float tl_, tr_;                                                                                          \
            const bool sl_ = slab(lmin_, lmax_, RAY, inv, TMAX_BOX, tl_);                                            \
            const bool sr_ = slab(rmin_, rmax_, RAY, inv, TMAX_BOX, tr_);                                            \
            const bool gl_ = (l_ & (REF_LEAF | REF_PAIR)) != REF_LEAF; /* Split nodes and leaf pairs carry a box */  \
            const bool gr_ = (r_ & (REF_LEAF | REF_PAIR)) != REF_LEAF;                                               \
            el_ = gl_ ? tl_ : 0.0f;                                                                                  \
            er_ = gr_ ? tr_ : 0.0f;                                                                                  \
            /* hl = nonempty && (!boxed || (slab && (vol || !(t > cull)))), cull being +inf until a leaf is hit --   */ \
            /* combined as wave masks in scalar registers: written on bools the compiler turns the expression into   */ \
            /* selects of 0/1 vectors (five vector instructions per child, ten per node step).                       */ \
            hl_ = lane_of(wave_mask(hl_) & (~wave_mask(gl_) | (wave_mask(sl_) & (wave_mask(lvol_) | ~wave_mask(tl_ > best_cull(best)))))); \
            hr_ = lane_of(wave_mask(hr_) & (~wave_mask(gr_) | (wave_mask(sr_) & (wave_mask(rvol_) | ~wave_mask(tr_ > best_cull(best)))))); \
        } else {                                                                                                     \
            if (hl_ && (l_ & (REF_LEAF | REF_PAIR)) != REF_LEAF) {                                                   \
                hl_ = slab(lmin_, lmax_, RAY, inv, TMAX_BOX, el_);                                                   \
                if (!lvol_ && el_ > best_cull(best)) hl_ = false;                                                        \
            }                                                                                                        \
            if (hr_ && (r_ & (REF_LEAF | REF_PAIR)) != REF_LEAF) {                                                   \
                hr_ = slab(rmin_, rmax_, RAY, inv, TMAX_BOX, er_);                                                   \
                if (!rvol_ && er_ > best_cull(best)) hr_ = false;                                                        \
            }                                                                                                        \
        }                                                                                                            \
        RTR_LS(14, hl_ && hr_)                                                                                       \
        if (hl_ && hr_) {                                                                                            \
            bool sw_ = er_ < el_;                                                                                    \
            if (lvol_ && rvol_) sw_ = false; /* both subtrees draw: keep the reference's left-then-right order */    \
            const uint32_t second_ = sw_ ? l_ : r_;                                                                  \
            float esecond_ = sw_ ? el_ : er_;                                                                        \
            if (sw_ ? lvol_ : rvol_) esecond_ = 0.0f; /* never culled at pop time: no bound is below tmin */        \
            cur = sw_ ? r_ : l_;                                                                                     \
            if constexpr (stack_packed) {                                                                            \
                stk_ref[sp * 64 + lane] = stack_pack32(second_, esecond_);                                           \
            } else {                                                                                                 \
                stk_ref[sp * 64 + lane] = second_;                                                                   \
                stk_t[sp * 64 + lane] = stack_t_pack(esecond_);                                                      \
            }                                                                                                        \
            sp++;                                                                                                    \
        } else if (hl_) {                                                                                            \
            cur = l_;                                                                                                \
        } else if (hr_) {                                                                                            \
            cur = r_;                                                                                                \
        } else {                                                                                                     \
            pop_ = true;                                                                                             \
        }                                                                                                            \
    }
#define RTR_WALK_TRIP(RAY, TMAX_BOX, ON_EMPTY)                                                                           \
    {                                                                                                                    \
        RTR_LS1(1) RTR_LS(2, true)                                                                                       \
        bool pop_ = false;                                                                                               \
        /* (cur is never REF_NONE here: the callers test the root, a node step tests both children before descending) */ \
        const bool leaf0_ = (cur & REF_LEAF) != 0u;                                                                      \
        if (!leaf0_) {                                                                                                   \
            RTR_LS1(6) RTR_LS(5, true)                                                                                   \
            const float4 *nq_ = reinterpret_cast<const float4 *>(p.nodes + cur);                                         \
            /* the two refs are an 8-byte load unless the flags word is needed: a fourth 16-byte load costs the vector */ \
            /* L1 a fifth of the node gather's time, an 8-byte one nothing (see test_leaf)                             */ \
            float4 q0_ = nq_[0], q1_ = nq_[1], q2_ = nq_[2], q3_;                                                        \
            if constexpr (VOL != 0u) {                                                                                   \
                q3_ = nq_[3];                                                                                            \
                gather_fence(q0_, q1_, q2_, q3_);                                                                        \
            } else {                                                                                                     \
                const float2 refs_ = *reinterpret_cast<const float2 *>(nq_ + 3);                                         \
                gather_fence(q0_, q1_, q2_, refs_);                                                                      \
                q3_ = make_float4(refs_.x, refs_.y, 0.0f, 0.0f);                                                         \
            }                                                                                                            \
            RTR_NODE_STEP(RAY, TMAX_BOX, q0_, q1_, q2_, q3_)                                                             \
        }                                                                                                                \
        /* A lane whose node step has just descended into a leaf tests it in the same trip (and so does a lane that stood */ \
        /* at a leaf when the trip began): a leaf visit costs no trip of its own.  Same operations per lane in the same  */ \
        /* order; 5-6 % fewer wave trips' worth of time on C1 / C3 / C4 (round 2).                                       */ \
        if (!pop_ && (cur & REF_LEAF)) {                                                                                             \
            RTR_LS1(4) RTR_LS(3, true)                                                                                   \
            /* a REF_PAIR ref stands for Split(leaf L, leaf L+1), its box already tested where the ref was read */       \
            counts.leaves++;                                                                                             \
            test_leaf<VOL>(p, cur & ~REF_PAIR, RAY, inv, a, best, rng, stk_ref, sp, lane);                               \
            pop_ = !(cur & REF_PAIR);                                                                                    \
            cur = (cur & ~REF_PAIR) + 1u;                                                                                \
        }                                                                                                                \
        if (pop_) {                                                                                                      \
            RTR_LS1(8) RTR_LS(7, true)                                                                                   \
            bool got_ = false;                                                                                           \
            while (sp > 0) {                                                                                             \
                RTR_LS1(9) RTR_LS(10, true)                                                                              \
                sp--;                                                                                                    \
                uint32_t c_ = stk_ref[sp * 64 + lane];                                                                   \
                float e_;                                                                                                \
                if constexpr (stack_packed) { e_ = stack_t32(c_); c_ = stack_ref32(c_); }                                \
                else e_ = stack_t_unpack(stk_t[sp * 64 + lane]);                                                         \
                if (e_ > best_cull(best)) continue; /* its box is entered beyond the best hit (never for 0 entries) */   \
                cur = c_;                                                                                                \
                got_ = true;                                                                                             \
                break;                                                                                                   \
            }                                                                                                            \
            if (!got_) { ON_EMPTY; }                                                                                     \
        }                                                                                                                \
    }

// BvhNode::hit over the tree rooted at `root`, whose own box has already been tested by the caller: the wave-synchronous
// form (the wave returns when its slowest lane has finished).
template <uint32_t VOL, bool MESH>
RTR_DEV void traverse(const RenderParams &p, uint32_t root, const Ray &r, V3 inv, float a, float tmax_box,
                      Best &best, uint32_t *stk_ref, StackT *stk_t, uint32_t lane, Rng &rng) {
    uint32_t sp = 0;
    uint32_t cur = root;
    bool walking = true;
    constexpr bool stack_packed = false;
    WalkCounts counts;
    while (walking) RTR_WALK_TRIP(r, tmax_box, walking = false)
}

struct Hit {
    V3 point, normal;
    float t; // (read by k_probe_hits only: dead, hence free, in the render kernels)
    bool front_face;
    uint32_t material;
    uint32_t slot; // sphere world with an LDS table: the winner's slot, which also addresses a copy of its material record
};

// HitRecord::new (ray.rs:39-54)
RTR_DEV void face_forward(Hit &h, const Ray &r, V3 outward) {
    h.front_face = dot(r.d, outward) < 0.0f;
    h.normal = h.front_face ? outward : neg(outward);
}

// Build the HitRecord of the winning leaf (objects.rs:45-48 | mesh.rs:185-193).
template <uint32_t VOL>
RTR_DEV void finish_leaf(const RenderParams &p, const Best &best, const Ray &r, Hit &h) {
    uint32_t L = best.leaf & REF_INDEX;
    const float4 *cold = reinterpret_cast<const float4 *>(p.leaf_cold + L);
    h.t = best.t;
    h.point = add(r.o, scale(r.d, best.t));
    float4 c0 = cold[0];
    h.material = __float_as_uint(c0.w);
    if (best.leaf & REF_TRI) {
        float4 c1 = cold[1], c2 = cold[2], c3 = cold[3];
        V3 n0 = mk(c0.x, c0.y, c0.z), n1 = mk(c1.x, c1.y, c1.z), n2 = mk(c2.x, c2.y, c2.z), fn = mk(c3.x, c3.y, c3.z);
        // the winner's barycentrics (mesh.rs:163-184), by Triangle::hit's own operations on its own operands: the values the
        // walk computed when it accepted this triangle
        const float4 *hot = reinterpret_cast<const float4 *>(p.leaf_hot + L);
        const float4 q0 = hot[0], q1 = hot[1];
        const float q2x = reinterpret_cast<const float *>(hot)[8];
        float tt, u, v;
        (void)triangle_t(mk(q0.x, q0.y, q0.z), mk(q0.w, q1.x, q1.y), mk(q1.z, q1.w, q2x), r, tt, u, v);
        V3 n = add(add(scale(n1, u), scale(n2, v)), scale(n0, 1.0f - u - v));
        // mesh.rs:187: normal.angle(face) > 0 with cgmath angle = atan2(|n x f|, n.f)
        V3 cr = cross(n, fn);
        float mag = rt_sqrt(dot(cr, cr));
        float dd = dot(n, fn);
        bool use_face = (mag > 0.0f) || (mag == 0.0f && (dd < 0.0f || (dd == 0.0f && __builtin_signbit(dd))));
        if (use_face) n = fn;
        face_forward(h, r, n);
    } else if (VOL && (best.leaf & REF_VOL)) {
        face_forward(h, r, mk(1.0f, 0.0f, 0.0f)); // volume.rs:67: the arbitrary constant normal
    } else {
        const float4 *hot = reinterpret_cast<const float4 *>(p.leaf_hot + L);
        const float4 q0 = hot[0];
        const float radius = reinterpret_cast<const float *>(hot)[4];
        V3 outward = divs(sub(h.point, mk(q0.x, q0.y, q0.z)), radius);
        face_forward(h, r, outward);
    }
}

// Linear world of NS <= 8 spheres: centre and r^2 of every sphere live in scalar registers for the whole
// kernel (loaded once, wave-uniform); the winner's radius / material come from a 32-B-per-sphere LDS table.
template <uint32_t NS>
struct SphereRegs {
    float cx[NS ? NS : 1], cy[NS ? NS : 1], cz[NS ? NS : 1], r2[NS ? NS : 1];
};

// world.hit (render.rs:38).  Returns false on miss.
template <uint32_t WORLD, uint32_t NS, uint32_t VOL, bool MESH>
RTR_DEV bool world_hit(const RenderParams &p, const Ray &r, Hit &h, uint32_t *stk_ref, StackT *stk_t, uint32_t lane,
                       const SphereRegs<NS> &sr, const uint32_t *lds_spheres, Rng &rng) {
    const float a = dot(r.d, r.d); // objects.rs:25, hoisted per ray
    if constexpr (WORLD == WORLD_LINEAR_SPHERES && NS > 0) {
        // scene.rs:141-157, fully unrolled over scalar-register sphere constants.  (A two-pass variant -- candidate
        // bits from the discriminants, then a per-lane loop over candidates only -- was measured 1 % slower:
        // +10 VGPRs and an LDS fetch per candidate outweigh the skipped root blocks.)
        float closest = __builtin_inff();
        int best = -1;
        // Deferred division.  Sphere::hit divides each root's numerator n = -half_b -/+ sqrt(disc) by a = |d|^2 only
        // to COMPARE the quotient with tmin and with closest_so_far (objects.rs:38-44); all roots of one ray share
        // the denominator, so the comparisons can be made on the numerators:
        //     q = RN(n / a) < tmin        is certain when  n < tmin*a*(1 - 2^-20),   its negation when  n > tmin*a*(1 + 2^-20)
        //     RN(nb / a) < RN(n / a)      is certain when  n > nb*(1 + 2^-20),       its negation when  n < nb*(1 - 2^-20)
        // (a > 0; one rounding moves a quotient by 2^-24 relative, the products carry 2^-23, the margin is 2^-20).
        // closest_so_far is kept as the numerator nb of the accepted root and divided once, for the winner: the
        // very value the reference stored.  A lane for which ANY needed comparison falls inside a margin (or whose a
        // is not a normal number in [2^-60, 2^60], or that meets a NaN) repeats the whole scan with the reference's
        // divisions below, so every decision is the reference's; the band is ~1e-6 wide, the repeat is rare.
        {
            const bool safe = a >= 0x1p-60f && a <= 0x1p60f;
            const float ta = kTMin * a;
            const float ta_lo = ta * 0x1.ffffep-1f, ta_hi = ta * 0x1.00001p+0f;
            float nb = __builtin_inff(), nb_lo = __builtin_inff(), nb_hi = __builtin_inff();
            bool amb = !safe;
            // Straight-line code: every sphere's root block runs for every lane, its outcome applied by selects.  A conditional
            // block per sphere (round 2) skipped the square root for the lanes that miss a sphere, but each block and each
            // decision inside it cost scalar instructions (exec-mask save / branch / restore: ~100 per trip for the five
            // spheres against ~50 mask operations now), and scalar instructions are not free on this kernel (60 extra ones per
            // trip, +20 %, cost +9.7 % of the frame time; 60 extra vector instructions, +10 %, cost +6.0 %); without the branches
            // the compiler also interleaves the five bodies, which fills the wait states between a compare and the select that
            // reads it (272 s_nop in the loop before, 8 in this region now).  13.77 -> 13.63 ms on the headline frame.
#pragma unroll
            for (uint32_t i = 0; i < NS; i++) {
                V3 oc = sub(r.o, mk(sr.cx[i], sr.cy[i], sr.cz[i]));
                float half_b = dot(oc, r.d);
                float cc = dot(oc, oc) - sr.r2[i];
                float disc = half_b * half_b - a * cc;
                const bool on = !(disc < 0.0f); // objects.rs:31-33 (a NaN discriminant goes on, as in the reference)
                bool fast;                       // disc in the range where rt_sqrt_core is sqrtf (else: the exact scan below)
                const float sq = rt_sqrt_core(disc, fast);
                const float n1 = -half_b - sq, n2 = -half_b + sq;
                const bool c1 = n1 < ta_lo;      // first root certainly < tmin: the second root decides
                const float n = c1 ? n2 : n1;
                const bool none = (c1 && n2 < ta_lo) || n > nb_hi; // certainly < tmin, or certainly beyond closest_so_far
                const bool ok = n > ta_hi && n < nb_lo;            // certainly accepted
                const bool take = on && fast && !none && ok;
                amb = amb || (on && (!fast || (!none && !ok)));
                nb = take ? n : nb;
                nb_lo = nb * 0x1.ffffep-1f; // (of +inf while nothing is accepted: +inf)
                nb_hi = nb * 0x1.00001p+0f;
                best = take ? (int)i : best;
            }
            RTR_LS(25, amb)
            if (!amb) {
                if (best >= 0) closest = nb / a;
            } else {
                closest = __builtin_inff();
                best = -1;
                for (uint32_t i = 0; i < NS; i++) { // the reference's scan, divisions and all
                    const float4 w = *reinterpret_cast<const float4 *>(lds_spheres + i * 8);
                    float t;
                    if (sphere_t(mk(w.x, w.y, w.z), __uint_as_float(lds_spheres[i * 8 + 4]), r, a, closest, t)) {
                        closest = t;
                        best = (int)i;
                    }
                }
            }
        }
        if (best < 0) return false;
        RTR_LS(12, true)
        const float4 w = *reinterpret_cast<const float4 *>(lds_spheres + best * 8); // centre, radius
        h.material = lds_spheres[best * 8 + 5];
        h.slot = (uint32_t)best;
        h.t = closest;
        h.point = add(r.o, scale(r.d, closest));
        const V3 pc = sub(h.point, mk(w.x, w.y, w.z));
        V3 outward;
        // (p - c) / r, objects.rs:46: three quotients over one radius, whose reciprocal was prepared at staging
        const float r1 = __uint_as_float(lds_spheres[best * 8 + 6]);
        if (lds_spheres[best * 8 + 7] != 0u && shared_div_numerators_ok(pc))
            outward = mk(shared_div(pc.x, w.w, r1), shared_div(pc.y, w.w, r1), shared_div(pc.z, w.w, r1));
        else
            outward = divs(pc, w.w);
        face_forward(h, r, outward);
        return true;
    } else if constexpr (WORLD == WORLD_LINEAR_SPHERES) {
        // scene.rs:141-157: ordered scan, closest_so_far shrinks, later object wins exact ties
        float closest = __builtin_inff();
        int best = -1;
        for (uint32_t i = 0; i < p.n_objects; i++) {
            const DevLinSphere &s = p.lin_spheres[i]; // wave-uniform address: scalar loads
            float t;
            if (sphere_t(mk(s.center[0], s.center[1], s.center[2]), s.radius2, r, a, closest, t)) {
                closest = t;
                best = (int)i;
            }
        }
        if (best < 0) return false;
        const DevLinSphere &s = p.lin_spheres[best];
        h.t = closest;
        h.point = add(r.o, scale(r.d, closest));
        V3 outward = divs(sub(h.point, mk(s.center[0], s.center[1], s.center[2])), s.radius);
        face_forward(h, r, outward);
        h.material = s.material;
        return true;
    } else {
        const V3 inv = mk(1.0f / r.d.x, 1.0f / r.d.y, 1.0f / r.d.z); // aabb.rs:86,103,120, hoisted per ray
        Best best;
        best_reset(best);
        if constexpr (WORLD == WORLD_BVH) {
            uint32_t root = p.root_ref;
            bool go = root != REF_NONE;
            if (go && (root & (REF_LEAF | REF_PAIR)) != REF_LEAF) {
                float e;
                go = slab(p.root_box, p.root_box + 3, r, inv, __builtin_inff(), e);
            }
            if (go) traverse<VOL, MESH>(p, root, r, inv, a, __builtin_inff(), best, stk_ref, stk_t, lane, rng);
        } else {
            // scene.rs:141-157 with meshes: object.hit(ray, tmin, closest_so_far); a mesh returns the nearest
            // triangle of the boxes that pass AABB::hit(.., closest_so_far) and Triangle::hit ignores tmax
            // (mesh.rs:151), so a mesh hit replaces the result unconditionally.
            float closest = __builtin_inff();
            for (uint32_t i = 0; i < p.n_objects; i++) {
                const DevObject &o = p.objects[i];
                uint32_t ref = o.ref;
                if (ref == REF_NONE) continue;
                if ((ref & REF_LEAF) && !(ref & REF_TRI)) {
                    const float4 *hot = reinterpret_cast<const float4 *>(p.leaf_hot + (ref & REF_INDEX));
                    float4 q0 = hot[0], q1 = hot[1], q2 = hot[2];
                    float t;
                    const bool got = (VOL && (ref & REF_VOL)) ? volume_leaf_t<VOL>(p, q0, q1, q2, r, inv, a, closest, rng, stk_ref, 0u, lane, t)
                                                     : sphere_t(mk(q0.x, q0.y, q0.z), q0.w, r, a, closest, t);
                    if (got) {
                        closest = t; best_take(best, t, ref);
                    }
                } else {
                    Best local;
                    best_reset(local);
                    bool go = true;
                    if ((ref & (REF_LEAF | REF_PAIR)) != REF_LEAF) {
                        float e;
                        go = slab(o.box, o.box + 3, r, inv, closest, e);
                    }
                    if (go) traverse<VOL, true>(p, ref, r, inv, a, closest, local, stk_ref, stk_t, lane, rng);
                    if (local.leaf != REF_NONE) {
                        closest = local.t;
                        best = local;
                    }
                }
            }
        }
        if (best.leaf == REF_NONE) return false;
        finish_leaf<VOL>(p, best, r, h);
        return true;
    }
}

// The launch parameters as the kernel argument segment holds them (RenderParams is the kernels' only argument: offset 0),
// behind a pointer the compiler cannot see through.  The per-ticket code of the integrate kernels reads its fields -- ticket
// counter, frame size, pixel order, carry and output pointers -- through it, so that they are (scalar-)loaded where a
// ticket begins or ends instead of being kept in scalar registers across the segment loop: that loop already needs the
// 96-register budget for the sphere constants and the camera, and every value spilled from it comes back as a
// v_readlane in the loop (+27 of them cost the headline kernel 15 %).
typedef const __attribute__((address_space(4))) RenderParams *KParams;
RTR_DEV const RenderParams &field(const RenderParams &p) { return p; }
RTR_DEV const __attribute__((address_space(4))) RenderParams &field(KParams p) { return *p; }
RTR_DEV KParams kernarg_params() {
    KParams k = (KParams)__builtin_amdgcn_kernarg_segment_ptr();
    asm volatile("" : "+s"(k));
    return k;
}

// --- shading -----------------------------------------------------------------------------------------
RTR_DEV bool near_zero(V3 v) { // math.rs:6-9
    const float eps = 1.0e-8f;
    return fabsf(v.x) < eps && fabsf(v.y) < eps && fabsf(v.z) < eps;
}
RTR_DEV V3 reflect(V3 v, V3 n) { return sub(v, scale(n, 2.0f * dot(v, n))); } // math.rs:12-14
RTR_DEV V3 refract(V3 uv, V3 n, float eta) {                                    // math.rs:16-21
    float cos_theta = fminf(dot(n, neg(uv)), 1.0f);
    V3 out_perp = scale(add(uv, scale(n, cos_theta)), eta);
    V3 out_parallel = scale(n, -rt_sqrt(fabsf(1.0f - dot(out_perp, out_perp))));
    return add(out_perp, out_parallel);
}
RTR_DEV float reflectance(float cosine, float idx) { // materials.rs:103-108; powi(x,5) = x*((x*x)*(x*x))
    float r0 = (1.0f - idx) / (1.0f + idx);
    r0 = r0 * r0;
    float x = 1.0f - cosine;
    float x2 = x * x;
    float x4 = x2 * x2;
    return r0 + (1.0f - r0) * (x * x4);
}

RTR_DEV float reflectance_r0(float cosine, float r0) { // the same with r0 = ((1 - idx) / (1 + idx))^2 given
    float x = 1.0f - cosine;
    float x2 = x * x;
    float x4 = x2 * x2;
    return r0 + (1.0f - r0) * (x * x4);
}

// Material records: from the LDS copy when the kernel staged one (lds_mats != nullptr), else from HBM/L1.
RTR_DEV void material_load(const RenderParams &p, const uint32_t *lds_mats, uint32_t id, float4 &q0, float4 &q1) {
    if (lds_mats) {
        const float4 *m = reinterpret_cast<const float4 *>(lds_mats + id * 8u);
        q0 = m[0]; q1 = m[1];
    } else {
        const float4 *m = reinterpret_cast<const float4 *>(p.materials + id);
        q0 = m[0]; q1 = m[1];
    }
}
RTR_DEV V3 material_rgb(const RenderParams &p, const uint32_t *lds_mats, uint32_t id) {
    if (id == kChainClay) return mk(0.8f, 0.8f, 0.8f);
    float4 q;
    if (lds_mats) q = *reinterpret_cast<const float4 *>(lds_mats + id * 8u);
    else q = *reinterpret_cast<const float4 *>(p.materials + id);
    return mk(q.x, q.y, q.z);
}

// SkyMap::sample's texel (backgrounds.rs:84-92, math.rs:24-29): index into the row-major image, row 0 = top
template <class P> // P: the launch parameters by reference or by kernarg pointer
RTR_DEV unsigned long long sky_texel_index(const P &pp, V3 dir) {
    const uint32_t sky_w = field(pp).sky_w, sky_h = field(pp).sky_h;
    V3 d = normalize(dir);
    float sx = poly_acos(-d.y);
    float sy = poly_atan2(-d.z, d.x) + kPi;
    float u = sx / kPi;
    float v = sy / (2.0f * kPi);
    // `(v * W) as usize % W` (backgrounds.rs:88-89).  v and u lie in [0, 1], so the product is below W except when the
    // coordinate is exactly 1: the 64-bit remainder -- a software division, ~150 instructions, twice per escaping path --
    // is taken only then (or for a NaN / out-of-range direction); the value is the same either way.
    const float fx = v * (float)sky_w, fy = u * (float)sky_h;
    uint32_t xi, yi;
    if (fx >= 0.0f && fx < (float)sky_w) xi = (uint32_t)fx; // (uint32_t)fx < W: no remainder to take
    else xi = (uint32_t)(sat_usize(fx) % sky_w);
    if (fy >= 0.0f && fy < (float)sky_h) yi = (uint32_t)fy;
    else yi = (uint32_t)(sat_usize(fy) % sky_h);
    const unsigned long long x = xi, y = (unsigned long long)sky_h - 1ull - yi;
    return x + y * sky_w;
}
// Option "sky_filter" = "bilinear" (SURVEY.md section 8f N2): the four texels around the direction instead of the nearest
// one -- longitude wraps, latitude clamps at the poles.  An extension, NOT what SkyMap::sample does; the oracle restates the
// same arithmetic (or_set_sky_filter) so that it can be checked bit for bit.
RTR_DEV V3 sky_bilinear(KParams p, V3 dir) {
    V3 d = normalize(dir);
    float sx = poly_acos(-d.y);
    float sy = poly_atan2(-d.z, d.x) + kPi;
    float u = sx / kPi;
    float v = sy / (2.0f * kPi);
    float fx = v * (float)p->sky_w - 0.5f, fy = u * (float)p->sky_h - 0.5f;
    if (!(fx == fx)) fx = 0.0f;
    if (!(fy == fy)) fy = 0.0f;
    const float x0f = floorf(fx), r0f = floorf(fy);
    const float tx = fx - x0f, ty = fy - r0f;
    const long long w = (long long)p->sky_w, hh = (long long)p->sky_h;
    long long x0 = (long long)x0f, r0 = (long long)r0f;
    long long x1 = x0 + 1, r1 = r0 + 1;
    x0 = ((x0 % w) + w) % w; x1 = ((x1 % w) + w) % w;
    r0 = r0 < 0 ? 0 : (r0 > hh - 1 ? hh - 1 : r0); r1 = r1 < 0 ? 0 : (r1 > hh - 1 ? hh - 1 : r1);
    const long long y0 = hh - 1 - r0, y1 = hh - 1 - r1;
    const float4 *sky = reinterpret_cast<const float4 *>(p->sky);
    const float4 t00 = sky[x0 + y0 * w], t01 = sky[x1 + y0 * w], t10 = sky[x0 + y1 * w], t11 = sky[x1 + y1 * w];
    const float ux = 1.0f - tx, uy = 1.0f - ty;
    const V3 a = add(scale(mk(t00.x, t00.y, t00.z), ux), scale(mk(t01.x, t01.y, t01.z), tx));
    const V3 b = add(scale(mk(t10.x, t10.y, t10.z), ux), scale(mk(t11.x, t11.y, t11.z), tx));
    return add(scale(a, uy), scale(b, ty));
}

// Background::sample (backgrounds.rs)
RTR_DEV V3 background(const RenderParams &p, const Ray &r) {
    if (p.bg_kind == 0u) return ld3(p.bg_a); // backgrounds.rs:25-30
    // (the other kinds read their parameters through the kernarg pointer: loaded where used, not held in scalar registers
    // across the segment loops of scenes that never come here)
    const KParams k = kernarg_params();
    if (p.bg_kind == 1u) {                   // backgrounds.rs:44-49, ray.rs:19-26
        float t = 0.5f * (normalize(r.d).y + 1.0f);
        return add(scale(mk(k->bg_a[0], k->bg_a[1], k->bg_a[2]), 1.0f - t), scale(mk(k->bg_b[0], k->bg_b[1], k->bg_b[2]), t));
    }
    if (k->sky_bilinear) return sky_bilinear(k, r.d); // opt-in (option "sky_filter"), not reference behaviour
    // backgrounds.rs:84-92, math.rs:24-29: nearest texel of an equirectangular image
    const float4 texel = reinterpret_cast<const float4 *>(k->sky)[sky_texel_index(k, r.d)];
    return mk(texel.x, texel.y, texel.z);
}

// The attenuation chain of a path: the ids of the materials whose attenuation multiplies the terminal radiance, in
// bounce order.  LdsChain: a per-lane column of the wave's LDS, [entry][lane] (conflict-free).  GlobalChain: one u32 per
// (entry, path) in HBM, [entry][path], for the streaming kernels whose paths outlive a kernel.
template <bool IDS_POSSIBLE = true> // false: the launch is known to keep attenuations (max_ray_depth <= 16): no id form, no test
struct LdsChain { // the sphere world's: LDS to spare, nothing to gain from the register form (measured: -2 ... -3 %)
    // Holds the attenuations THEMSELVES (three floats per bounce, [entry][channel][lane]), not material ids: the replay
    // is then one LDS round trip per entry instead of two dependent ones (id, then the material's colour), in the block of
    // the loop that a quarter of the wave's cycles go to although it is a seventh of its instructions.
    static constexpr bool kStoresRgb = true;
    float *base;
    uint32_t lane, n;
    bool ids; // wave-uniform (RenderParams::chain_bits == 0xFFFFFFFF): fall back to one id per entry when three words per
              // entry would not fit LDS (max_ray_depth above 16; the reference allows 255)
    RTR_DEV void init(uint32_t *lds_base, uint32_t lane_id, uint32_t bits) {
        base = reinterpret_cast<float *>(lds_base); lane = lane_id; n = 0u; ids = IDS_POSSIBLE && bits == 0xFFFFFFFFu;
    }
    RTR_DEV void reset() { n = 0u; }
    RTR_DEV void push(uint32_t id, V3 rgb) {
        if (IDS_POSSIBLE && ids) {
            reinterpret_cast<uint32_t *>(base)[n * 64u + lane] = id;
        } else {
            float *e = base + n * 192u + lane;
            e[0] = rgb.x; e[64] = rgb.y; e[128] = rgb.z;
        }
        n++;
    }
    RTR_DEV V3 pop_rgb(const RenderParams &p, const uint32_t *lds_mats) {
        n--;
        if (IDS_POSSIBLE && ids) return material_rgb(p, lds_mats, reinterpret_cast<const uint32_t *>(base)[n * 64u + lane]);
        const float *e = base + n * 192u + lane;
        return mk(e[0], e[64], e[128]);
    }
};
struct PackChain { // the tree worlds'
    static constexpr bool kStoresRgb = false;
    uint32_t *base;
    uint32_t lane, n;
    // width != 0 (a wave-uniform choice of the host: RenderParams::chain_bits): the ids do not go through LDS at all but
    // sit in `packed`, `width` bits each -- possible whenever max_ray_depth ids of the scene's material table fit 32
    // bits (gui.rs scene: 3 x 3 bits).  The wave then needs no chain region in LDS, and LDS is what bounds the tree
    // kernels' occupancy (house.obj: 6.8 -> 6.0 KB per wave, C4 +5 %).  All-ones is the Clay id.
    uint32_t packed, width;
    RTR_DEV void init(uint32_t *lds_base, uint32_t lane_id, uint32_t bits) { base = lds_base; lane = lane_id; width = bits; n = 0u; packed = 0u; }
    RTR_DEV void reset() { n = 0u; packed = 0u; }
    RTR_DEV void push(uint32_t id, V3) {
        if (width) packed |= (id == kChainClay ? (1u << width) - 1u : id) << (n * width);
        else base[n * 64u + lane] = id;
        n++;
    }
    RTR_DEV V3 pop_rgb(const RenderParams &p, const uint32_t *lds_mats) { return material_rgb(p, lds_mats, pop()); }
    RTR_DEV uint32_t at(uint32_t j) const {
        if (width) {
            const uint32_t mask = (1u << width) - 1u, code = (packed >> (j * width)) & mask;
            return code == mask ? kChainClay : code;
        }
        return base[j * 64u + lane];
    }
    RTR_DEV uint32_t pop() {
        n--;
        const uint32_t id = at(n);
        if (n == 0u) packed = 0u;
        return id;
    }
};
struct GlobalChain {
    static constexpr bool kStoresRgb = false;
    uint32_t *base;
    uint64_t stride; // paths per entry row
    uint32_t path, n;
    RTR_DEV void push(uint32_t id, V3) { base[(uint64_t)n * stride + path] = id; n++; }
    RTR_DEV uint32_t pop() { n--; return base[(uint64_t)n * stride + path]; }
    RTR_DEV V3 pop_rgb(const RenderParams &p, const uint32_t *lds_mats) { return material_rgb(p, lds_mats, pop()); }
    RTR_DEV void reset() { n = 0u; }
};

// f32::clamp(0, clamp_indirect) per channel with NaN staying NaN (render.rs:95-97)
RTR_DEV V3 clamp_radiance(const RenderParams &p, V3 term) {
    const float hi = p.clamp_indirect;
    if (term.x < 0.0f) term.x = 0.0f;
    if (term.x > hi) term.x = hi;
    if (term.y < 0.0f) term.y = 0.0f;
    if (term.y > hi) term.y = hi;
    if (term.z < 0.0f) term.z = 0.0f;
    if (term.z > hi) term.z = hi;
    return term;
}
// attenuation * (attenuation * (... * terminal)) (render.rs:52), then the clamp: the radiance one finished sample adds
// to its pixel
template <class Chain>
RTR_DEV V3 sample_radiance(const RenderParams &p, const uint32_t *lds_mats, Chain &chain, V3 term) {
    if constexpr (Chain::kStoresRgb) {
        // two entries per round: their reads are independent, so the replay costs one LDS round trip per PAIR of bounces
        while (chain.n >= 2u) {
            const V3 inner = chain.pop_rgb(p, lds_mats), outer = chain.pop_rgb(p, lds_mats);
            term = mul(outer, mul(inner, term));
        }
    }
    while (chain.n > 0u) {
        term = mul(chain.pop_rgb(p, lds_mats), term);
    }
    return clamp_radiance(p, term);
}

// The part of trace_ray (render.rs:38-58) after world.hit: background on a miss, else the material's scatter.
// Returns `done` (the path ended: `term` is its terminal radiance) or false with the next segment in `ray`.
// PLAIN: the instantiation for launches the host knows to be RenderMode::Full over a UniformBackground with no MixMaterial
// and no Isotropic in the material table (benchmark.json): the wave-uniform mode / background / mix tests -- scalar
// compares and branches on every trip -- and the code behind them leave the loop.  Same operations for every path it serves.
template <bool PLAIN = false, class Chain>
RTR_DEV bool shade_segment(const RenderParams &p, bool was_hit, const Hit &h, Ray &ray, Rng &rng, const uint32_t *lds_mats,
                           Chain &chain, uint32_t &depth, V3 &term, bool *defer_sky = nullptr, const uint32_t *slot_mats = nullptr) {
    bool done = false;
    uint32_t kind = 7u, mid = 0u; // 7: no material to scatter from (inactive lane, miss, or a mode that returns at once)
    float param = 0.0f;
    V3 rgb = mk(0.8f, 0.8f, 0.8f);
    float4 mq1 = make_float4(0.0f, 0.0f, 0.0f, 0.0f); // the material record's second half (Dielectric: 1/ir and the two r0^2)
    RTR_LS(17, !was_hit)
    {
        if (!was_hit) {
            // defer_sky: the caller evaluates SkyMap::sample later, from the direction (k_render_bvh_sp -> k_sum_samples)
            if constexpr (PLAIN) term = ld3(p.bg_a); // UniformBackground::sample, backgrounds.rs:25-30
            else if (defer_sky && p.bg_kind == 2u && !p.sky_bilinear) { *defer_sky = true; term = ray.d; }
            else term = background(p, ray);
            done = true;
        } else if (PLAIN) { // Full mode, no MixMaterial: the hit sphere's material record, addressed by the sphere's slot
            mid = h.material;
            const float4 *m = reinterpret_cast<const float4 *>(slot_mats + h.slot * 8u);
            const float4 q0 = m[0], q1 = m[1];
            kind = __float_as_uint(q1.x);
            rgb = mk(q0.x, q0.y, q0.z);
            param = q0.w;
            mq1 = q1;
        } else if (p.mode == 3u) { // RenderMode::Normal, render.rs:42-45
            V3 n = scale(add(normalize(h.normal), mk(1.0f, 1.0f, 1.0f)), 0.5f);
            term = n;
            done = true;
        } else if (p.mode == 2u) { // RenderMode::Random, render.rs:46-48
            term = mk(0.0f, 0.0f, 0.0f);
            done = true;
        } else if (p.mode == 1u) { // RenderMode::Clay: Lambertian(0.8) (render.rs:41)
            kind = 0u; mid = kChainClay;
        } else {
            // MixMaterial::scatter (materials.rs:148-157): xi >= factor -> first
            mid = h.material;
            for (bool first = true;; first = false) {
                float4 q0, q1;
                if (slot_mats && first) { // the hit sphere's material record, addressed by the sphere's slot: this read does not
                    const float4 *m = reinterpret_cast<const float4 *>(slot_mats + h.slot * 8u); // wait for the material id
                    q0 = m[0]; q1 = m[1];
                } else material_load(p, lds_mats, mid, q0, q1);
                kind = __float_as_uint(q1.x);
                rgb = mk(q0.x, q0.y, q0.z);
                param = q0.w;
                mq1 = q1;
                if (kind != 4u) break;
                mid = (rng_unit(rng) >= param) ? __float_as_uint(q1.y) : __float_as_uint(q1.z);
            }
        }
    }
    // Lambertian, Metal and Isotropic all start from one rejection-sampled point of the unit ball
    // (sample.rs:43-51) and it is their first draw after the mix walk, so the sampler is run ONCE for all
    // three kinds (one divergent block per segment instead of one per material kind).
    V3 sv = mk(0.0f, 0.0f, 0.0f);
    RTR_LS(13, kind == 0u) RTR_LS(14, kind == 1u) RTR_LS(15, kind == 2u) RTR_LS(16, kind == 3u)
    const bool need_ball = kind == 0u || kind == 1u || (!PLAIN && kind == 5u);
    float m2 = 1.0f;
    if (need_ball) {
#ifdef RTR_X_LANESTAT
        RTR_LS1(20)
        uint32_t bs = rng_u32(rng);
        do {
            RTR_LS(19, true) RTR_LS1(18)
            sv.x = ball_coord(bs); sv.y = ball_coord(bs); sv.z = ball_coord(bs);
            m2 = dot(sv, sv);
        } while (!(m2 <= 1.0f));
#else
        sv = sample_unit_ball(rng, m2);
#endif
    }
    if (kind != 7u) {
        bool scattered = true;
        V3 ndir = h.normal;
        // One pass of `v * (1 / |v|)` (cgmath normalize) for every kind that normalises a vector first: the ball sample of
        // Lambertian and Metal (sample_unit_sphere_surface, sample.rs:66-70) and the ray direction of Dielectric
        // (materials.rs:116).  Same operations per lane; one block for ~34 lanes instead of two for 25 and 9.
        V3 unit = mk(0.0f, 0.0f, 0.0f);
        if (kind == 0u || kind == 1u || kind == 3u) {
            const bool di = kind == 3u;
            const V3 x = di ? ray.d : sv;
            const float x2 = di ? dot(ray.d, ray.d) : m2;
            unit = scale(x, 1.0f / rt_sqrt(x2));
            if (!di) sv = unit;
        }
        if (kind == 0u) { // Lambertian, materials.rs:31-41
            ndir = add(h.normal, sv);
            if (near_zero(ndir)) ndir = h.normal;
        } else if (kind == 1u) { // Metal, materials.rs:58-70
            V3 reflected = normalize(reflect(ray.d, h.normal));
            ndir = add(reflected, scale(sv, param));
            if (!(dot(ndir, h.normal) > 0.0f)) { scattered = false; term = mk(0.0f, 0.0f, 0.0f); }
        } else if (kind == 2u) { // Emission, materials.rs:86-91
            scattered = false; term = rgb;
        } else if (kind == 3u) { // Dielectric, materials.rs:111-138
            // 1/ir and the two Schlick r0^2 values are constants of the material, divided once on the host
            // (rtr_scene.cpp) into the record's spare words
            const float4 dq1 = mq1;
            const float ratio = h.front_face ? dq1.y : param;
            const float r0sq = h.front_face ? dq1.z : dq1.w;
            float cos_theta = fminf(dot(h.normal, neg(unit)), 1.0f);
            float sin_theta = rt_sqrt(1.0f - cos_theta * cos_theta);
            bool cannot_refract = (ratio * sin_theta) > 1.0f;
            bool refl = cannot_refract;
            if (!refl) refl = reflectance_r0(cos_theta, r0sq) > rng_unit(rng); // no draw on TIR (short-circuit ||)
            ndir = refl ? reflect(unit, h.normal) : refract(unit, h.normal, ratio);
            mid = 0xFFFFFFFEu; // attenuation (1,1,1): x*1 == x exactly, nothing to replay
        } else { // Isotropic, volume.rs:80-88: the un-normalised ball sample is the direction
            ndir = sv;
        }
        if (scattered) {
            if (mid != 0xFFFFFFFEu) {
                chain.push(mid, rgb);
            }
            ray.o = h.point;
            ray.d = ndir;
            depth++;
            done = depth >= p.max_depth; // render.rs:34-36
            term = mk(0.0f, 0.0f, 0.0f);
        } else {
            done = true;
        }
    }
    return done;
}

// Tickets and sample chunks.  A launch of k_render / k_render_bvh is one workgroup (one wave) per TICKET; a ticket is
// (chunk c, group g): the 8x8 tile g of the launch and the c-th of p.chunks consecutive pieces of every pixel's sample
// range, tickets numbered chunk-major.  A workgroup takes its ticket from a counter in device memory when it starts
// running -- not from its block index.  Chunk c of a tile starts from the running sums chunk c-1 left in p.carry_rgba and
// adds its samples in the same order, so every pixel sees exactly the additions of the unchunked loop: bit-identical for
// any chunk count.  The hand-over is a per-tile counter: written with release, read with acquire (agent scope).
// Forward progress does not depend on dispatch order (HIP promises none): a workgroup waits only for the ticket (c-1, g),
// whose number is smaller than its own and which was therefore claimed EARLIER -- by a workgroup that is running or has
// finished (tickets are claimed by running workgroups only) and that itself waits, if at all, for a still smaller
// ticket; by induction every claimed ticket completes.  (Round 2 derived the ticket from blockIdx and relied on
// workgroups being dispatched in index order, with a bounded spin as the net.)  The wait stays bounded in wall time and a
// timeout sets the context's fault word, as a belt; it cannot fire on a healthy device.
// Why chunks at all: the tail of the launch then consists of short pieces instead of whole tiles (at 1080p the frame is
// only ~4.5 tiles per resident wave, and 11 % of the time was spent with the chip draining).
// Measured and not adopted (round 3, headline frame, all bit-identical): persistent workgroups --
// one per resident wave, each claiming tickets until none is left -- 15.1-15.3 ms against 14.05: the same instructions
// and wave-cycles, but 9 % fewer waves resident on average (in-kernel clocks: waves live 14.3 of the launch's 16.2 ms);
// and groups of 64 pixels sorted by a cost key from a 4-sample pre-pass instead of 8x8 tiles -- 16.9 ms: 6.6 % MORE
// wave-cycles in the segment loop (neighbouring pixels are the most coherent company a pixel can have) and a longer tail;
// and 256-thread workgroups whose four waves deal their 256 paths out anew by material kind between world.hit and the
// rest of trace_ray, through LDS (counting sort by wave ballots, a 10-word record per path there and back, three barriers
// per segment; commit 2466776) -- bit-identical, 21.8 ms: the lanes ARE fuller (0.48 -> 0.62 of VALU lane-cycles active)
// but the sort and exchange bookkeeping adds 197 lane-instructions to a segment's 316, and 40 % of the wave time is
// spent parked at barriers and LDS waits (22 % in this kernel).
// First sample (relative) of chunk c of n.  With taper k (0..7) the chunk lengths fall linearly from (8+k)/8 of the mean
// to (8-k)/8: what is left running when no tickets remain are pieces of the LAST chunk, so that one should be the smallest.
RTR_DEV uint32_t chunk_boundary(uint32_t count, uint32_t c, uint32_t n, uint32_t taper) {
    if (taper == 0u || n < 2u) return (uint32_t)((uint64_t)count * c / n);
    // cumulative weight of chunks 0..c-1 with weight_i = 8 + k - 2k*i/(n-1), 8n in total
    const uint64_t k = taper;
    const uint64_t num = (uint64_t)c * (8u + k) * (n - 1u) - k * (uint64_t)c * (c - 1u);
    return (uint32_t)((uint64_t)count * num / (8ull * n * (n - 1u)));
}

RTR_DEV void chunk_wait(KParams p, uint32_t tidx, uint32_t chunk) {
    // Bounded by wall time (s_memrealtime: 100 MHz, independent of the shader clock).  Giving up sets bit 0 of the
    // context's sticky fault word, which every synchronising entry point of the C ABI turns into RTR_ERR_HIP
    // (rtr_api.cpp check_fault).
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    while (__hip_atomic_load(p->chunk_flags + tidx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < chunk) {
        __builtin_amdgcn_s_sleep(16);
        if (__builtin_amdgcn_s_memrealtime() - t0 > p->chunk_timeout_ticks) {
            if (threadIdx.x == 0) atomicOr(p->fault, 1ull);
            break;
        }
    }
    asm volatile("" ::: "memory"); // the carry loads are issued after the counter has been seen (the loop's own vmcnt(0))
}
RTR_DEV void chunk_publish(KParams p, uint32_t tidx, uint32_t chunk, uint32_t lane) {
    // the wave's carry stores have completed before the counter moves (a one-wave workgroup's fence compiles to nothing)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
    if (lane == 0) __hip_atomic_store(p->chunk_flags + tidx, chunk + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// The hand-over's memory traffic is made of device-scope atomic accesses only -- the carry words and the counter are written
// and read with relaxed agent-scope atomics, which gfx950 performs at the device's coherence point (`sc1`: write-through
// stores, loads that do not trust the XCD's L2) -- ordered by the wave's own completion counter (`s_waitcnt vmcnt(0)`: a wait, no cache operation).  Round 2 / early round 3 used plain stores with __threadfence() + a
// release store, and an acquire load: on a GPU of eight L2s an agent-scope release writes back the XCD's whole L2
// (`buffer_wbl2 sc1`, twice per hand-over here) and an acquire invalidates it (`buffer_inv sc1`), for every workgroup of
// the launch, under everybody else's feet: chunk counts above four cost far more than their prologues (8 chunks: 15.2 ms
// against 13.2) and the frame itself paid for ~97 000 such pairs.
RTR_DEV V3 carry_load(KParams p, uint32_t pixel) {
    const float *c = p->carry_rgba + 4ull * pixel;
    V3 v;
    v.x = __hip_atomic_load(c + 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    v.y = __hip_atomic_load(c + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    v.z = __hip_atomic_load(c + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return v;
}
RTR_DEV void carry_store(KParams p, uint32_t pixel, V3 v) {
    float *c = p->carry_rgba + 4ull * pixel;
    __hip_atomic_store(c + 0, v.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(c + 1, v.y, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(c + 2, v.z, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// The workgroup's ticket (wave-uniform); false when the launch has more workgroups than tickets.  (Only the ticket's number
// stays live across the segment loop; chunk and tile are divided out of it again where the ticket ends.)
RTR_DEV void ticket_split(KParams p, uint32_t id, uint32_t &chunk, uint32_t &tidx) {
    const uint32_t per_chunk = p->tiles_launch;
    chunk = id / per_chunk;
    tidx = id - chunk * per_chunk;
}
RTR_DEV bool claim_ticket(KParams p, uint32_t lane, uint32_t &id) {
    id = 0u;
    if (lane == 0u) id = atomicAdd(p->work_counter, 1u);
    id = __builtin_amdgcn_readfirstlane(id);
    return id < p->work_blocks;
}

// calculate_pixel's per-sample head (render.rs:80-90) and Camera::get_ray (camera.rs:19-24): the launch's wave-uniform
// constants, and the start of sample `si` of the pixel whose stream hash is `hp`.
constexpr uint32_t kSampleKeyStep = 0x9E3779B9u; // odd: sample index -> key is one-to-one modulo 2^32
struct Primary {
    V3 o, h, v, llc;
    SharedDiv div_w, div_h; // (x + xi) / (W - 1), (y + xi') / (H - 1): wave-uniform denominators in [1, 2^31] (else the plain
    bool uv_fast;           // division), numerators 0 or >= 2^-23 -- inside shared_div's range
    float wm1, hm1;
    RTR_DEV void init(const RenderParams &p) {
        o = ld3(p.cam_origin); h = ld3(p.cam_horizontal); v = ld3(p.cam_vertical); llc = ld3(p.cam_llc);
        wm1 = p.width_m1; hm1 = p.height_m1;
        div_w = shared_div_prepare(wm1); div_h = shared_div_prepare(hm1);
        uv_fast = div_w.ok && div_h.ok;
    }
    RTR_DEV void start(uint32_t hp, uint32_t si, float fx, float fy, Rng &rng, Ray &ray) const {
        start_key(hp + si * kSampleKeyStep, fx, fy, rng, ray);
    }
    // ... given the sample's stream key hp + si * kSampleKeyStep (k_render steps the key from sample to sample)
    RTR_DEV void start_key(uint32_t key, float fx, float fy, Rng &rng, Ray &ray) const {
        rng.state = hash32(key);
        float uu = fx + rng_unit_inclusive(rng); // render.rs:88-89
        float vv = fy + rng_unit_inclusive(rng);
        if (uv_fast) {
            uu = shared_div(uu, div_w.d, div_w.r1);
            vv = shared_div(vv, div_h.d, div_h.r1);
        } else {
            uu = uu / wm1;
            vv = vv / hm1;
        }
        ray.o = o; // camera.rs:19-24
        ray.d = sub(sub(add(llc, scale(h, uu)), scale(v, vv)), o);
    }
};

// render.rs:103-110 for one pixel: sum / spp, then sqrt gamma and RGBA8 (color.rs:34-41); and / or the bare sum (sharded paths)
RTR_DEV void store_pixel(V3 acc, uint32_t spp, uint32_t pixel, float *sum_rgb, uint8_t *rgba8, float *mean_rgb) {
    if (sum_rgb) { // three floats per pixel, packed: what the multi-GPU reduce ships (no padding word on the wire)
        sum_rgb[3ull * pixel + 0] = acc.x; sum_rgb[3ull * pixel + 1] = acc.y; sum_rgb[3ull * pixel + 2] = acc.z;
    }
    if (rgba8 || mean_rgb) {
        const float n = (float)spp;
        const V3 mean = mk(acc.x / n, acc.y / n, acc.z / n);
        if (mean_rgb) {
            mean_rgb[3ull * pixel + 0] = mean.x;
            mean_rgb[3ull * pixel + 1] = mean.y;
            mean_rgb[3ull * pixel + 2] = mean.z;
        }
        if (rgba8) {
            const uint32_t r8 = sat_u8(sqrtf(mean.x) * 255.0f), g8 = sat_u8(sqrtf(mean.y) * 255.0f), b8 = sat_u8(sqrtf(mean.z) * 255.0f);
            reinterpret_cast<uint32_t *>(rgba8)[pixel] = r8 | (g8 << 8) | (b8 << 16) | (255u << 24);
        }
    }
}
// the material table's copy in LDS (one wave per workgroup: the caller's LDS wait is the only barrier needed)
RTR_DEV void stage_materials(const RenderParams &p, uint32_t *lds_mat_base, uint32_t lane) {
    for (uint32_t i = lane; i < p.lds_materials * 8u; i += 64u) lds_mat_base[i] = reinterpret_cast<const uint32_t *>(p.materials)[i];
}
// one atomic per wave: the ray segments its lanes traced
RTR_DEV void count_segments(uint32_t nseg, uint32_t lane) {
    unsigned long long total = nseg;
    for (int off = 32; off > 0; off >>= 1) total += __shfl_xor(total, off, 64);
    if (lane == 0 && total) atomicAdd(kernarg_params()->counter, total);
}
// A ticket's 64 pixels, one per lane: the 8x8 tile `tidx` of this shard.  render.rs:76-77: index = y * W + x.
RTR_DEV bool ticket_pixel(KParams p, uint32_t tidx, uint32_t lane, uint32_t &pixel, float &fx, float &fy) {
    const uint32_t width = p->width;
    const uint32_t tile = tidx * p->n_shards + p->shard, tiles_x = p->tiles_x;
    const uint32_t tx = tile % tiles_x, ty = tile / tiles_x;
    const uint32_t x = tx * 8u + (lane & 7u), y = ty * 8u + (lane >> 3);
    pixel = y * width + x;
    fx = (float)x; fy = (float)y;
    return tile < p->tiles_total && x < width && y < p->height;
}

// What a lane needs to start its pixel's share of a ticket: the pixel's stream constants, the sample range, the running sum.
struct TicketLane {
    uint32_t pixel;                   // 0xFFFFFFFF: the lane has no pixel in this ticket
    uint32_t hp, inc, s_first, s_end; // s_end: one past the lane's last sample (= s_first for a lane without a pixel)
    float fx, fy;
    V3 acc;
};
RTR_DEV TicketLane ticket_begin(uint32_t id, uint32_t lane) {
    const KParams k = kernarg_params();
    uint32_t chunk, tidx;
    ticket_split(k, id, chunk, tidx);
    TicketLane t;
    const bool inside = ticket_pixel(k, tidx, lane, t.pixel, t.fx, t.fy);
    // per-pixel stream constants
    t.hp = hash32(t.pixel ^ k->seed_key0); // the two keys are hashes of the whole 64-bit seed (rtr_api.cpp)
    t.inc = (hash32(t.hp + k->seed_key1) << 1) | 1u;
    // samples [sample_begin, sample_begin + sample_count) of this pixel; the random stream is keyed by the
    // absolute sample index, so a frame split by samples over several launches / GPUs draws the same numbers
    const uint32_t begin = k->sample_begin, count = k->sample_count, chunks = k->chunks, taper = k->chunk_taper;
    uint32_t s_last;
    if (k->chunk_table_on) { // the host's table (rtr_api.cpp): up to eight chunks
        t.s_first = begin + k->chunk_first[chunk];
        s_last = begin + k->chunk_first[chunk + 1u];
    } else {
        t.s_first = begin + chunk_boundary(count, chunk, chunks, taper);
        s_last = begin + chunk_boundary(count, chunk + 1u, chunks, taper);
    }
    // (per lane on purpose: the wave-uniform form -- lanes off the frame's edge starting at s_last -- saves a register
    // and measured 0.8 % slower)
    t.s_end = (inside && k->max_depth > 0u) ? s_last : t.s_first;
    t.acc = mk(0.0f, 0.0f, 0.0f);
    if (chunk > 0u) { // continue from the sums the previous chunk of this group left behind
        chunk_wait(k, tidx, chunk);
        if (inside) {
            t.acc = carry_load(k, t.pixel);
        }
    }
    if (!inside) t.pixel = 0xFFFFFFFFu;
    return t;
}
// ... and to end it: hand the running sum to the next chunk, or finish the pixel
RTR_DEV void ticket_end(uint32_t id, uint32_t lane, uint32_t pixel, V3 acc) {
    const KParams k = kernarg_params();
    uint32_t chunk, tidx;
    ticket_split(k, id, chunk, tidx);
    const bool inside = pixel != 0xFFFFFFFFu;
    if (chunk + 1u < k->chunks) {
        if (inside) carry_store(k, pixel, acc);
        chunk_publish(k, tidx, chunk, lane);
    } else if (inside) {
        store_pixel(acc, k->spp, pixel, k->sum_rgb, k->rgba8, k->mean_rgb);
    }
}

// One lane = one pixel of the workgroup's ticket.  LDS (dynamic, 16-byte aligned pieces):
//   [chain: max_depth x 64 u32][stack refs: cap x 64][stack t: cap x 64][spheres: NS x 8][materials: lds_mats x 8]
template <uint32_t WORLD, uint32_t NS, uint32_t VOL, bool MESH, bool PLAIN = false>
// Register budget of the sphere world with up to five spheres in scalar registers (the headline): 8 waves per SIMD -- 64
// vector registers, and 80 scalar ones (a wave's SGPRs come in 16-register granules out of 800 per SIMD: 96 admit 7
// waves, 80 admit 8).  Round 2 could not afford it (64 VGPRs only with 51 spilled SGPRs: 14.8 ms against 14.05 at 7 waves);
// with the per-ticket parameters read through the kernarg pointer the 8-wave build spills 13 SGPRs, none in the segment
// loop: 13.79 ms against 13.95 at 7 waves.  More spheres in registers, the generic sphere loop and the tree worlds keep
// the 7-wave / 96-SGPR budget (unasked the compiler takes 77 VGPRs = 6 waves: 15.4 ms on the headline frame).
__attribute__((amdgpu_num_sgpr(96))) // (the cap of the 7-wave instantiations; an 8-wave launch bound implies 80)
__global__ __launch_bounds__(64, WORLD == WORLD_LINEAR_SPHERES ? (NS >= 1u && NS <= 5u ? 8 : 7) : 1) void k_render(const RenderParams p) {
    extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
    const uint32_t lane = threadIdx.x;
#ifdef RTR_X_LIFE // diagnostic build only (scripts/wave_life.py): how long the launch's workgroups live, and how long before their loop
    const unsigned long long life_t0 = __builtin_amdgcn_s_memrealtime();
    unsigned long long life_t1 = life_t0;
#endif
#ifdef RTR_X_LANESTAT
    if (lane < 32u) rtr_lsc[lane] = 0u;
    __builtin_amdgcn_s_waitcnt(0);
    __builtin_amdgcn_wave_barrier();
#endif
    typename std::conditional<WORLD == WORLD_LINEAR_SPHERES, LdsChain<!PLAIN>, PackChain>::type chain;
    chain.init(lds, lane, p.chain_bits);
    uint32_t *stk_ref = lds + p.chain_lds * 64u; // (the sphere world's chain_lds counts three words per entry)
    StackT *stk_t = reinterpret_cast<StackT *>(stk_ref + p.stack_cap * 64u);
    uint32_t *lds_spheres = reinterpret_cast<uint32_t *>(stk_t + p.stack_cap * 64u);
    uint32_t *lds_mat_base = lds_spheres + NS * 8u;
    const uint32_t *lds_mats = p.lds_materials ? lds_mat_base : nullptr;
    // stage the small tables (one wave per workgroup: no barrier needed beyond the LDS wait)
    SphereRegs<NS> sr;
    if constexpr (NS > 0) {
#pragma unroll
        for (uint32_t i = 0; i < NS; i++) {
            const DevLinSphere &sp = p.lin_spheres[i]; // uniform -> scalar loads, kept in SGPRs
            sr.cx[i] = sp.center[0]; sr.cy[i] = sp.center[1]; sr.cz[i] = sp.center[2]; sr.r2[i] = sp.radius2;
        }
        for (uint32_t i = lane; i < NS * 8u; i += 64u) lds_spheres[i] = reinterpret_cast<const uint32_t *>(p.lin_spheres)[i];
        __builtin_amdgcn_s_waitcnt(0);
        __builtin_amdgcn_wave_barrier();
        if (lane < NS) { // the two pad words of the LDS copy: the radius' prepared reciprocal and its validity
            const SharedDiv sd = shared_div_prepare(__uint_as_float(lds_spheres[lane * 8u + 3u]));
            lds_spheres[lane * 8u + 6u] = __float_as_uint(sd.r1);
            lds_spheres[lane * 8u + 7u] = sd.ok ? 1u : 0u;
        }
    }
    if (p.lds_materials) stage_materials(p, lds_mat_base, lane);
    // sphere world: a copy of each sphere's material record next to the sphere table, addressed by the sphere's slot, so
    // that the record of the hit sphere's material can be fetched together with the sphere's own record
    uint32_t *slot_mats = nullptr;
    if constexpr (NS > 0) {
        slot_mats = lds_mat_base + p.lds_materials * 8u;
        for (uint32_t i = lane; i < NS * 8u; i += 64u)
            slot_mats[i] = reinterpret_cast<const uint32_t *>(p.materials)[p.lin_spheres[i >> 3].material * 8u + (i & 7u)];
    }
    __builtin_amdgcn_s_waitcnt(0); // vmcnt(0) lgkmcnt(0): the tables are in LDS before any lane reads them
    __builtin_amdgcn_wave_barrier();

    Primary cam;
    cam.init(p);
    uint32_t nseg = 0;
#ifdef RTR_X_STAMPS // diagnostic build only: wave-cycle shares of the loop's sections -> counter ring slots 1..5
    unsigned long long st_acc[5] = {0, 0, 0, 0, 0};
    unsigned long long st_last = __builtin_amdgcn_s_memtime();
#define RTR_STAMP(k) { __builtin_amdgcn_sched_barrier(0); unsigned long long st_now = __builtin_amdgcn_s_memtime(); \
        __builtin_amdgcn_s_waitcnt(0xC07F); st_acc[k] += st_now - st_last; st_last = st_now; __builtin_amdgcn_sched_barrier(0); }
#else
#define RTR_STAMP(k)
#endif

    uint32_t ticket;
    if (claim_ticket(kernarg_params(), lane, ticket)) {
        const TicketLane tk = ticket_begin(ticket, lane);
        const uint32_t pixel = tk.pixel, hp = tk.hp, spp = tk.s_end;
        const float fx = tk.fx, fy = tk.fy;
        Rng rng;
        rng.inc = tk.inc;
        rng.state = 0u;
        V3 acc = tk.acc;
        // the lane's sample counter lives as the sample's stream key (one add per sample instead of a quarter-rate multiply)
        uint32_t key = hp + tk.s_first * kSampleKeyStep, depth = 0;
        const uint32_t key_end = hp + spp * kSampleKeyStep;
        Ray ray;
        ray.o = cam.o; ray.d = cam.o;
        if (key != key_end) { cam.start_key(key, fx, fy, rng, ray); chain.reset(); }
#ifdef RTR_X_LIFE
        __builtin_amdgcn_s_waitcnt(0);
        life_t1 = __builtin_amdgcn_s_memrealtime();
#endif

        while (key != key_end) {
            RTR_STAMP(4)
            RTR_LS(1, true) RTR_LS1(0)
            // ---- trace_ray body (render.rs:32-59): one segment ----
            Hit h;
            nseg++;
            V3 term;
            const bool was_hit = world_hit<WORLD, NS, VOL, MESH>(p, ray, h, stk_ref, stk_t, lane, sr, lds_spheres, rng);
            RTR_STAMP(0)
            const bool done = shade_segment<PLAIN>(p, was_hit, h, ray, rng, lds_mats, chain, depth, term, nullptr, slot_mats);
            RTR_STAMP(1)
            RTR_LS(21, done)
            if (done) {
                RTR_LS1(22)
                acc = add(acc, sample_radiance(p, lds_mats, chain, term)); // render.rs:99
                key += kSampleKeyStep;
                if (key != key_end) { cam.start_key(key, fx, fy, rng, ray); depth = 0; chain.reset(); }
            }
            RTR_STAMP(3)
        }

        ticket_end(ticket, lane, pixel, acc);
    }
#ifdef RTR_X_LIFE
    if (p.spp > 1u) { // scripts/wave_life.py makes 3 one-sample launches first, so slots counter-1..-3 exist
        __builtin_amdgcn_s_waitcnt(0);
        const unsigned long long life_t2 = __builtin_amdgcn_s_memrealtime();
        if (lane == 0) {
            atomicAdd(p.counter - 1, life_t2 - life_t0); // lifetimes, in 10 ns
            atomicAdd(p.counter - 2, life_t1 - life_t0); // before the loop
            atomicAdd(p.counter - 3, 1ull);              // workgroups
        }
    }
#endif
#ifdef RTR_X_STAMPS
    if (lane == 0 && p.spp > 1u) // scripts/stamps.py makes 5 one-sample launches first, so slots counter-1..-5 exist
        for (int k = 0; k < 5; k++) atomicAdd(p.counter - 1 - k, st_acc[k]);
#endif
#ifdef RTR_X_LANESTAT
    __builtin_amdgcn_s_waitcnt(0);
    __builtin_amdgcn_wave_barrier();
    if (lane < 32u && p.spp > 1u) // scripts/lanestat.py makes 32 one-sample launches first, so slots counter-1..-32 exist
        atomicAdd(p.counter - 1 - lane, (unsigned long long)rtr_lsc[lane]);
#endif
    count_segments(nseg, lane);
}

// ---- WORLD_BVH with a resumable walk -------------------------------------------------------------------------------
// In k_render a wave leaves world.hit only when its slowest lane has finished its walk: rays that need 10 trips wait
// for rays that need 60 (C4: 19 % of VALU lane-cycles active).  Here the walk is a per-lane state machine inside the
// main loop: lanes take traversal trips until no more than 5/8 (kQuorumMesh) of the wave's live lanes are still
// walking, then the lanes whose walk has ended shade, scatter or regenerate together and rejoin with a fresh walk, while
// the late walkers simply keep their state.  Every lane executes exactly the operations it executes in k_render, in the
// same order (its random stream included), so the frame is bit-identical; only the interleaving between lanes changes.
// Walk while more than 5/8 of the live lanes are walking in mesh scenes, 2/8 in sphere scenes, whose shading rounds are
// short next to a trip (102 spheres: 0 / 1 / 2 / 3 / 4 eighths 13.8 / 14.3 / 14.4 / 14.1 / 13.9 Grays/s; meshes: 4/8 ... 6/8
// within 2 %, 5/8 best on house.obj, 4/8 on monkey.obj); kWalkUnroll trips per quorum check: C3 +15 %, C4 +5 % over one
// (2, 3, 4, 6, 8 measured: flat from 3 on).
constexpr uint32_t kQuorumMesh = 5u, kQuorumSpheres = 2u, kQuorumDen = 8u;
constexpr uint32_t kWalkUnroll = 4u;

// The head of world_hit<WORLD_BVH> for a lane that starts a segment: per-ray constants, the root's box.  Returns whether
// the lane has a walk to take (the state is in cur / sp / best / inv / a).
// (Measured and not adopted, round 3: the ROOT's node step done here, on the root's record carried in the launch parameters
// and read by scalar loads -- every walk starts at the root, so it takes one trip out of the 13.3 lane-trips of a segment
// on house.obj -- C1 / C3 / C4 70.23 / 83.80 / 70.07 ms against 70.26 / 83.98 / 70.05 without: the trip it saves is the
// cheapest one, all lanes gathering one record, and its work moves into the latency-exposed shading round.)
RTR_DEV bool begin_walk(const RenderParams &p, const Ray &ray, V3 &inv, float &a, Best &best, uint32_t &cur, uint32_t &sp) {
    a = dot(ray.d, ray.d);                                        // objects.rs:25, hoisted per ray
    inv = mk(1.0f / ray.d.x, 1.0f / ray.d.y, 1.0f / ray.d.z);     // aabb.rs:86,103,120, hoisted per ray
    best_reset(best);
    (void)p;
    const KParams k = kernarg_params(); // (the root's ref and box: seven scalars loaded here instead of living across the walk)
    const uint32_t root = k->root_ref;
    bool go = root != REF_NONE;
    if (go && (root & (REF_LEAF | REF_PAIR)) != REF_LEAF) {
        float e;
        const float bmin[3] = {k->root_box[0], k->root_box[1], k->root_box[2]}, bmax[3] = {k->root_box[3], k->root_box[4], k->root_box[5]};
        go = slab(bmin, bmax, ray, inv, __builtin_inff(), e);
    }
    cur = root; sp = 0u;
    return go;
}

// (80 VGPRs, 6 waves per SIMD: the 7-wave budget (72) spills since the cull bound joined the walk's state -- C4 +5 %, C3 +11 % at 6)
template <uint32_t VOL, bool MESH>
__attribute__((amdgpu_num_sgpr(96)))
__global__ __launch_bounds__(64, 6) void k_render_bvh(const RenderParams p) {
    extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
    const uint32_t lane = threadIdx.x;
    PackChain chain;
    chain.init(lds, lane, p.chain_bits);
    uint32_t *stk_ref = lds + p.chain_lds * 64u;
    StackT *stk_t = reinterpret_cast<StackT *>(stk_ref + p.stack_cap * 64u);
    uint32_t *lds_mat_base = reinterpret_cast<uint32_t *>(stk_t + p.stack_cap * 64u);
    const uint32_t *lds_mats = p.lds_materials ? lds_mat_base : nullptr;
    if (p.lds_materials) stage_materials(p, lds_mat_base, lane);
    __builtin_amdgcn_s_waitcnt(0);
    __builtin_amdgcn_wave_barrier();
    Primary cam;
    cam.init(p);
    uint32_t nseg = 0;
    constexpr bool stack_packed = false;
    WalkCounts counts;

    uint32_t ticket;
    if (claim_ticket(kernarg_params(), lane, ticket)) { // tickets, chunks and their hand-over: as in k_render
        const TicketLane tk = ticket_begin(ticket, lane);
        const uint32_t pixel = tk.pixel, hp = tk.hp, spp = tk.s_end;
        const float fx = tk.fx, fy = tk.fy;
        Rng rng;
        rng.inc = tk.inc;
        rng.state = 0u;
        V3 acc = tk.acc;
        uint32_t s = tk.s_first, depth = 0;
        Ray ray;
        ray.o = cam.o; ray.d = cam.o;
        // the walk's per-lane state (world_hit + traverse of k_render, unrolled into the loop below)
        uint32_t cur = REF_NONE, sp = 0;
        Best best;
        best_reset(best);
        V3 inv = mk(0.0f, 0.0f, 0.0f);
        float a = 0.0f;
        bool walking = false;
        bool alive = s < spp;
        if (alive) {
            cam.start(hp, s, fx, fy, rng, ray); chain.reset();
            nseg++;
            walking = begin_walk(p, ray, inv, a, best, cur, sp);
        }
        for (;;) {
            const uint64_t alive_mask = __ballot(alive);
            if (alive_mask == 0ull) break;
            const uint32_t quorum = (uint32_t)__popcll(alive_mask) * kQuorumMesh / kQuorumDen;
            // ---- walk: one trip per iteration for every lane still walking (RTR_WALK_TRIP: the one trip body) ----
            // (Taking only the kind of step -- node or leaf -- that more walkers wait for was measured 0 ... -10 %: with both
            // bodies in one trip their gathers overlap.)
            while ((uint32_t)__popcll(__ballot(walking)) > quorum) {
#pragma unroll
                for (uint32_t rep = 0; rep < kWalkUnroll; rep++)
                    if (walking) RTR_WALK_TRIP(ray, __builtin_inff(), walking = false)
            }
            // ---- the lanes whose walk has ended: the rest of trace_ray, then the next segment or the next sample ----
            if (alive && !walking) {
                Hit h;
                const bool was_hit = best.leaf != REF_NONE;
                if (was_hit) finish_leaf<VOL>(p, best, ray, h);
                V3 term;
                const bool done = shade_segment(p, was_hit, h, ray, rng, lds_mats, chain, depth, term);
                if (done) {
                    acc = add(acc, sample_radiance(p, lds_mats, chain, term)); // render.rs:99
                    s++;
                    if (s < spp) { cam.start(hp, s, fx, fy, rng, ray); depth = 0; chain.reset(); }
                    else alive = false;
                }
                if (alive) {
                    nseg++;
                    walking = begin_walk(p, ray, inv, a, best, cur, sp);
                }
            }
        }
        ticket_end(ticket, lane, pixel, acc);
    }
    count_segments(nseg, lane);
}

// ---- WORLD_BVH, samples instead of pixels ------------------------------------------------------------------------
// k_render_bvh ties a lane to one pixel because the pixel's samples are SUMMED in order -- but they need not be RENDERED
// in order.  Here the launch is one workgroup (one wave) per wave the device keeps resident; a wave claims blocks of
// (tile, a few consecutive samples) from a counter in device memory and its lanes take the block's (sample, pixel) pairs
// as they finish (ballot + mbcnt hand-out); each sample's clamped radiance is written to p.samples[sample][pixel], and
// k_sum_samples adds a pixel's samples in index order afterwards (the same additions as calculate_pixel's loop).  No lane
// waits for its tile's slowest pixel, no wave for a slower one, and nothing is handed over between workgroups.
// Register budget (round 3: two-register best hit, per-block parameters read through the kernarg pointer -- the 7-wave
// build no longer spills a vector register): mesh scenes 8 waves per SIMD (64 VGPRs, 9 spilled outside the walk: C3 / C4
// +1.6 / +1.5 % over 7), sphere scenes 7 (72 VGPRs; 8 waves: -0.9 %), the volume kernels 6.
// COUNT: the instantiation behind option "traversal_counters" -- the same kernel, which also adds up its lanes' node steps,
// leaf tests and SkyMap lookups (rtr_stats): a template flag, so that the counters cost the product kernel nothing.
template <uint32_t VOL, bool MESH, bool PACKED, bool COUNT = false>
__attribute__((amdgpu_num_sgpr(128)))
__global__ __launch_bounds__(64, VOL == 0u ? (MESH ? 8 : 7) : 6) void k_render_bvh_sp(const RenderParams p) {
    extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
    const uint32_t lane = threadIdx.x;
#ifdef RTR_X_LANESTAT
    if (lane < 32u) rtr_lsc[lane] = 0u;
    __builtin_amdgcn_s_waitcnt(0);
    __builtin_amdgcn_wave_barrier();
#endif
    PackChain chain;
    chain.init(lds, lane, p.chain_bits);
    uint32_t *stk_ref = lds + p.chain_lds * 64u;
    StackT *stk_t = reinterpret_cast<StackT *>(stk_ref + p.stack_cap * 64u);
    uint32_t *lds_mat_base = PACKED ? stk_ref + p.stack_cap * 64u : reinterpret_cast<uint32_t *>(stk_t + p.stack_cap * 64u);
    const uint32_t *lds_mats = p.lds_materials ? lds_mat_base : nullptr;
    if (p.lds_materials) stage_materials(p, lds_mat_base, lane);
    __builtin_amdgcn_s_waitcnt(0);
    __builtin_amdgcn_wave_barrier();

    // Work is claimed, not assigned: the launch is one workgroup per wave the device keeps resident, and a wave whose lanes
    // ask for work takes the next block -- one tile, samples_per_block consecutive samples: 64 (sample, pixel) pairs per
    // sample, sample-major, so the lanes that start together start on one sample's coherent primary rays -- from a
    // counter in device memory (one returning atomic per block: ~28 per microsecond at 1080p, a third of what a counter
    // word sustains).  With a fixed share per wave (round 2's first form: tile x sample range) every one of ~550 000
    // waves paid its own set-up and its own ragged end; now those are paid once per resident wave.
    uint32_t tx = 0u, ty = 0u;                         // the current block's tile (wave-uniform)
    Primary cam;
    cam.init(p);
    // the lane's current work item: one sample of one pixel of a tile
    uint32_t hp = 0;
    float fx = 0.0f, fy = 0.0f;
    uint32_t slot = 0; // where the sample's radiance goes: [sample - sample_begin][pixel] (a batch has fewer than 2^32 pairs)
    Rng rng;
    rng.inc = 1u; rng.state = 0u;
    uint32_t nseg = 0;
    uint32_t s_first = 0u, n_items = 0u, next_item = 0u; // the current block: its first sample, its pairs, the next unclaimed one
    bool pool_empty = false;                             // (all wave-uniform)
    uint32_t depth = 0;
    Ray ray;
    ray.o = cam.o; ray.d = cam.o;

    // the walk's per-lane state (world_hit + traverse of k_render, unrolled into the loop below)
    uint32_t cur = REF_NONE, sp = 0;
    Best best;
    best_reset(best);
    V3 inv = mk(0.0f, 0.0f, 0.0f);
    float a = 0.0f;
    // Lane states are wave masks in scalar registers (walking_m, alive_m, need_m), changed only where the wave is
    // converged: as per-lane bools carried round the loops they live in vector registers as bytes, re-tested and copied
    // on every trip.
    uint64_t walking_m = 0ull;
    constexpr bool stack_packed = PACKED; // one-word stack entries (RenderParams::stack_packed, set for small trees)
    WalkCounts counts;
    auto begin_segment = [&]() -> bool { nseg++; return begin_walk(p, ray, inv, a, best, cur, sp); }; // true = the lane has a walk to take
    uint64_t alive_m = 0ull;
    uint64_t need_m = wave_mask(true); // the lanes that want the workgroup's next (sample, pixel) pair
    const uint32_t n_pixels = p.width * p.height;
    auto fetch_work = [&]() { // hand the next items to the lanes that ask, in lane order; wave-uniform control flow
        bool began = false;
        while (need_m != 0ull) {
            if (next_item >= n_items) { // the block is used up: claim the next one (wave-uniform)
                if (pool_empty) { need_m = 0ull; break; } // nothing is left: the asking lanes retire
                const KParams k = kernarg_params(); // (the block's parameters are loaded here, not held across the walk)
                uint32_t id = 0u;
                if (lane == 0u) id = atomicAdd(k->work_counter, 1u);
                id = __builtin_amdgcn_readfirstlane(id);
                if (id >= k->work_blocks) { pool_empty = true; need_m = 0ull; break; }
                const uint32_t per_group = k->tiles_launch, tiles_x = k->tiles_x, per_block = k->samples_per_block, count = k->sample_count;
                const uint32_t group = id / per_group, tile = (id - group * per_group) * k->n_shards + k->shard;
                tx = tile % tiles_x; ty = tile / tiles_x;
                const uint32_t s0 = group * per_block;
                s_first = k->sample_begin + s0;
                n_items = 64u * (count - s0 < per_block ? count - s0 : per_block);
                next_item = 0u;
            }
            const uint64_t asking = need_m;
            const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(asking >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)asking, 0u));
            const uint32_t item = next_item + rank;
            next_item += (uint32_t)__popcll(asking);
            bool again = false, got = false;
            if (lane_of(asking)) {
                if (item >= n_items) {
                    again = true; // past the end of the block: ask again, from the next one
                } else {
                    const uint32_t pix = item & 63u, si = s_first + (item >> 6);
                    const uint32_t x = tx * 8u + (pix & 7u), y = ty * 8u + (pix >> 3);
                    if (x < p.width && y < p.height) {
                        const uint32_t pixel = y * p.width + x; // render.rs:76-77
                        fx = (float)x; fy = (float)y;
                        hp = hash32(pixel ^ p.seed_key0);
                        rng.inc = (hash32(hp + p.seed_key1) << 1) | 1u;
                        slot = (si - p.sample_begin) * n_pixels + pixel;
                        got = true;
                        cam.start(hp, si, fx, fy, rng, ray); depth = 0; chain.reset(); // render.rs:80-90, camera.rs:19-24
                        began = begin_segment();
                    } else {
                        again = true; // off the ragged edge: ask again
                    }
                }
            }
            need_m = wave_mask(again);
            alive_m |= wave_mask(got);
        }
        walking_m |= wave_mask(began);
    };
    fetch_work();

    for (;;) {
        if (alive_m == 0ull) break;
        const uint32_t quorum = (uint32_t)__popcll(alive_m) * (MESH ? kQuorumMesh : kQuorumSpheres) / kQuorumDen; // (see k_render_bvh)
        RTR_LS1(0) RTR_LS(15, lane_of(alive_m))
        // ---- walk: one trip per iteration for every lane still walking ----
        // (RTR_WALK_TRIP: the one trip body every walk expands in place.)
        while ((uint32_t)__popcll(walking_m) > quorum) {
#pragma unroll
            for (uint32_t rep = 0; rep < kWalkUnroll; rep++) {
                bool ended = false;
                if (lane_of(walking_m)) RTR_WALK_TRIP(ray, __builtin_inff(), ended = true)
                walking_m &= ~wave_mask(ended);
            }
        }
        // ---- the lanes whose walk has ended: the rest of trace_ray, then the next segment or the next sample ----
        bool began = false, finished = false;
        if (lane_of(alive_m & ~walking_m)) {
            RTR_LS1(11) RTR_LS(12, true)
            Hit h;
            const bool was_hit = best.leaf != REF_NONE;
            if (was_hit) finish_leaf<VOL>(p, best, ray, h);
            V3 term;
            bool sky = false;
            const bool done = shade_segment(p, was_hit, h, ray, rng, lds_mats, chain, depth, term, &sky);
            if (!was_hit && p.bg_kind == 2u) counts.sky++; // SkyMap::sample (evaluated here, or deferred to k_sum_samples)
            if (done) { // the sample's clamped radiance (render.rs:95-97) goes to memory; k_sum_samples adds them in order
                // A path that escapes to a SkyMap leaves its DIRECTION and its attenuation chain instead (up to 3 material
                // ids below 256, packed into the fourth word): k_sum_samples, whose lanes are all busy, does the
                // normalize / acos / atan2 / texel fetch / chain product / clamp -- the same operations on the same values
                // -- that would run here for the ~20 lanes of a shading round that escape, with an HBM-latency gather in
                // the middle of a loop of L2 hits.
                uint32_t packed = 0x80000000u | (chain.n << 24);
                bool fits = sky && chain.n <= 3u;
                for (uint32_t j = 0; fits && j < chain.n; j++) {
                    const uint32_t id = chain.at(j);
                    fits = id < 256u;
                    packed |= id << (8u * j);
                }
                if (fits) {
                    reinterpret_cast<float4 *>(p.samples)[slot] = make_float4(term.x, term.y, term.z, __uint_as_float(packed));
                    chain.reset();
                } else {
                    if (sky) term = background(p, ray);
                    const V3 rad = sample_radiance(p, lds_mats, chain, term);
                    reinterpret_cast<float4 *>(p.samples)[slot] = make_float4(rad.x, rad.y, rad.z, 0.0f);
                }
                finished = true;
            } else {
                began = begin_segment();
            }
        }
        walking_m |= wave_mask(began);
        need_m = wave_mask(finished);
        alive_m &= ~need_m;
        fetch_work();
    }

#ifdef RTR_X_LANESTAT
    __builtin_amdgcn_s_waitcnt(0);
    __builtin_amdgcn_wave_barrier();
    if (lane < 32u && p.spp > 1u) atomicAdd(p.counter - 1 - lane, (unsigned long long)rtr_lsc[lane]);
#endif
    count_segments(nseg, lane);
    if constexpr (COUNT) {
        unsigned long long t0 = counts.nodes, t1 = counts.leaves, t2 = counts.sky;
        for (int off = 32; off > 0; off >>= 1) { t0 += __shfl_xor(t0, off, 64); t1 += __shfl_xor(t1, off, 64); t2 += __shfl_xor(t2, off, 64); }
        if (lane == 0) {
            unsigned long long *out = kernarg_params()->walk_counters;
            atomicAdd(out, t0); atomicAdd(out + 1, t1); atomicAdd(out + 2, t2);
        }
    }
}


// ---- WORLD_BVH, streaming wavefront: path state in HBM, one kernel per stage ---------------------------------------
// The kernels above keep a path in one lane's registers from its first segment to its last, so a wave always holds a
// mix of lanes that walk the tree and lanes that shade, regenerate or wait for the quorum: on house.obj 34 of 64 lanes
// walk in an average trip (scripts/lanestat_tree.py), and the walk is nine tenths of the time.  Here the stages are
// separate kernels and the paths of a batch of samples live in HBM between them (BASELINE's streaming design: SURVEY.md
// section 8d prices it at ~100 B of path state per segment, which a tree walk of ~2500 lane-instructions can afford):
//   k_wf_generate            every (sample, pixel) of the batch: camera ray, random stream (render.rs:80-90)
//   per depth d = 0 .. max_ray_depth-1 (trace_ray's recursion level, render.rs:32-36):
//     k_wf_intersect<d>      world.hit for every path alive at depth d.  A lane owns nothing: when its walk ends it
//                            stores the hit and takes the next waiting path, so nearly every lane of every trip walks.
//     k_wf_shade<d>          the rest of trace_ray for those paths: background / material / scatter; a path that ends
//                            writes its sample's clamped radiance, one that goes on stores its next ray and moves to d+1.
//   k_sum_samples            adds each pixel's samples in index order (as for k_render_bvh_sp).
// A path is (sample, pixel); its random stream is keyed by both, and each path executes exactly the operations it
// executes in k_render, in the same order: frames are bit-identical to the other tree kernels'.
// No queues and no atomics: wf_stage[path] holds the depth a path waits at (255: finished).  A wave scans the stage
// bytes of its blocks of 64 paths (block-cyclic over the grid), drops the ids that match the kernel's depth into a small
// LDS ring and hands them to the lanes that ask.  Scanning costs one byte and a fraction of an instruction per path and
// depth; the alternative, compacted queues, needs one returning atomic per appended run, and a single counter word
// sustains ~88 of those per microsecond (MI355X_MICROARCH.md, dequeue) against ~13 M runs per 1080p frame.
constexpr uint32_t kWfRing = 128u; // ids waiting in LDS: a block adds at most 64 to at most 63 left over

struct WfScan {
    uint32_t next_block, head, tail; // wave-uniform
};
// Scan blocks until the ring holds `want` ids or the wave's blocks are exhausted.  Returns the number of ids in the ring.
RTR_DEV uint32_t wf_scan(const RenderParams &p, uint32_t depth, uint32_t lane, uint32_t *ring, WfScan &sc, uint32_t want) {
    while (sc.head - sc.tail < want && sc.next_block < p.wf_blocks) {
        const uint32_t id = sc.next_block * 64u + lane;
        sc.next_block += gridDim.x;
        const uint32_t stage = id < p.wf_paths ? (uint32_t)p.wf_stage[id] : 255u;
        const bool mine = stage == depth;
        const uint64_t m = __ballot(mine);
        if (mine) ring[(sc.head + __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u))) & (kWfRing - 1u)] = id;
        sc.head += (uint32_t)__popcll(m);
    }
    __builtin_amdgcn_s_waitcnt(0xc07f); // lgkmcnt(0): the ring entries are in LDS before any lane reads them
    __builtin_amdgcn_wave_barrier();
    return sc.head - sc.tail;
}

// render.rs:76-90 + camera.rs:19-24 for every (sample, pixel) of the batch.  grid = (tiles, samples of the batch), one
// wave = one 8x8 tile of one sample, so the paths of a block start out as coherent as k_render's waves.
__global__ __launch_bounds__(64) void k_wf_generate(const RenderParams p) {
    const uint32_t lane = threadIdx.x;
    const uint32_t tile = blockIdx.x, s_rel = blockIdx.y;
    const uint32_t id = (s_rel * p.tiles_total + tile) * 64u + lane;
    const uint32_t tx = tile % p.tiles_x, ty = tile / p.tiles_x;
    const uint32_t x = tx * 8u + (lane & 7u), y = ty * 8u + (lane >> 3);
    if (!(x < p.width && y < p.height)) { p.wf_stage[id] = 255u; return; }
    const uint32_t pixel = y * p.width + x; // render.rs:76-77
    const uint32_t hp = hash32(pixel ^ p.seed_key0);
    Rng rng;
    rng.inc = (hash32(hp + p.seed_key1) << 1) | 1u;
    rng.state = 0u;
    Primary cam;
    cam.init(p);
    Ray r;
    cam.start(hp, p.sample_begin + s_rel, (float)x, (float)y, rng, r); // render.rs:80-90, camera.rs:19-24
    float4 *ray = reinterpret_cast<float4 *>(p.wf_ray) + 2ull * id;
    ray[0] = make_float4(r.o.x, r.o.y, r.o.z, r.d.x);
    ray[1] = make_float4(r.d.y, r.d.z, 0.0f, 0.0f);
    reinterpret_cast<uint4 *>(p.wf_state)[id] = make_uint4(rng.state, rng.inc, 0u, s_rel * (p.width * p.height) + pixel);
    p.wf_stage[id] = 0u;
}

// world.hit (render.rs:38) for every path waiting at depth p.wf_depth.  LDS: [stack refs: cap x 64][stack t: cap x 64][ring]
template <uint32_t VOL, bool MESH>
__global__ __launch_bounds__(64, 6) void k_wf_intersect(const RenderParams p) {
    extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
    const uint32_t lane = threadIdx.x;
    uint32_t *stk_ref = lds;
    StackT *stk_t = reinterpret_cast<StackT *>(stk_ref + p.stack_cap * 64u);
    uint32_t *ring = reinterpret_cast<uint32_t *>(stk_t + p.stack_cap * 64u);
    const uint32_t depth = p.wf_depth;
    WfScan sc;
    sc.next_block = blockIdx.x; sc.head = 0u; sc.tail = 0u;

    // the walk's per-lane state
    uint32_t id = 0u, cur = REF_NONE, sp = 0u, nseg = 0u;
    Best best;
    best_reset(best);
    Ray ray;
    ray.o = mk(0.0f, 0.0f, 0.0f); ray.d = ray.o;
    V3 inv = ray.o;
    float a = 0.0f;
    Rng rng;
    rng.state = 0u; rng.inc = 1u;
    bool walking = false; // the lane's walk is in progress
    bool holding = false; // the lane's walk has ended and its hit is not stored yet
    WalkCounts counts;
    for (;;) {
        // ---- lanes without a path take the next waiting ones ----
        {
            const uint64_t asking = __ballot(!walking && !holding);
            if (asking != 0ull) {
                const uint32_t want = (uint32_t)__popcll(asking);
                const uint32_t have = wf_scan(p, depth, lane, ring, sc, want);
                const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(asking >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)asking, 0u));
                if (!walking && !holding && rank < have) {
                    id = ring[(sc.tail + rank) & (kWfRing - 1u)];
                    const float4 *rq = reinterpret_cast<const float4 *>(p.wf_ray) + 2ull * id;
                    const float4 r0 = rq[0], r1 = rq[1];
                    ray.o = mk(r0.x, r0.y, r0.z);
                    ray.d = mk(r0.w, r1.x, r1.y);
                    if constexpr (VOL) { // Volume::hit draws from the path's stream inside world.hit (volume.rs:57)
                        const uint2 rs = *reinterpret_cast<const uint2 *>(p.wf_state + 4ull * id);
                        rng.state = rs.x; rng.inc = rs.y;
                    }
                    nseg++;
                    walking = begin_walk(p, ray, inv, a, best, cur, sp); // the head of world_hit<WORLD_BVH>
                    holding = !walking;
                }
                sc.tail += want < have ? want : have;
            }
        }
        constexpr bool stack_packed = false;
        const uint64_t walkers = __ballot(walking);
        if (walkers == 0ull && __ballot(holding) == 0ull) break; // nothing in flight and nothing left to take
        // ---- walk: kWalkUnroll trips for every lane with a walk in progress ----
        if (walkers != 0ull) {
#pragma unroll
            for (uint32_t rep = 0; rep < kWalkUnroll; rep++)
                if (walking) RTR_WALK_TRIP(ray, __builtin_inff(), { walking = false; holding = true; })
        }
        // ---- finished walks are stored together, once enough of them wait (or nothing else is left to do) ----
        const uint64_t held = __ballot(holding);
        if (held != 0ull && ((uint32_t)__popcll(held) >= p.wf_flush || __ballot(walking) == 0ull)) {
            if (holding) {
                reinterpret_cast<float4 *>(p.wf_hit)[id] = make_float4(best.t, 0.0f, 0.0f, __uint_as_float(best.leaf));
                if constexpr (VOL) p.wf_state[4ull * id] = rng.state;
                holding = false;
            }
        }
    }
    count_segments(nseg, lane);
}

// The rest of trace_ray (render.rs:39-58) for every path waiting at depth p.wf_depth, full waves of them at a time.
// LDS: [ring][materials: lds_mats x 8]
template <uint32_t VOL>
__global__ __launch_bounds__(64) void k_wf_shade(const RenderParams p) {
    extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
    const uint32_t lane = threadIdx.x;
    uint32_t *ring = lds;
    uint32_t *lds_mat_base = lds + kWfRing;
    const uint32_t *lds_mats = p.lds_materials ? lds_mat_base : nullptr;
    if (p.lds_materials) stage_materials(p, lds_mat_base, lane);
    __builtin_amdgcn_s_waitcnt(0);
    __builtin_amdgcn_wave_barrier();
    const uint32_t depth0 = p.wf_depth;
    WfScan sc;
    sc.next_block = blockIdx.x; sc.head = 0u; sc.tail = 0u;
    for (;;) {
        const uint32_t have = wf_scan(p, depth0, lane, ring, sc, 64u);
        if (have == 0u) break;
        const uint32_t take = have < 64u ? have : 64u;
        if (lane < take) {
            const uint32_t id = ring[(sc.tail + lane) & (kWfRing - 1u)];
            const float4 *rq = reinterpret_cast<const float4 *>(p.wf_ray) + 2ull * id;
            const float4 r0 = rq[0], r1 = rq[1];
            const float4 hq = reinterpret_cast<const float4 *>(p.wf_hit)[id];
            const uint4 st = reinterpret_cast<const uint4 *>(p.wf_state)[id];
            Ray ray;
            ray.o = mk(r0.x, r0.y, r0.z);
            ray.d = mk(r0.w, r1.x, r1.y);
            Best best;
            best.t = hq.x; best.leaf = __float_as_uint(hq.w);
            Rng rng;
            rng.state = st.x; rng.inc = st.y;
            GlobalChain chain;
            chain.base = p.wf_chain; chain.stride = p.wf_paths; chain.path = id; chain.n = st.z;
            uint32_t depth = depth0;
            Hit h;
            const bool was_hit = best.leaf != REF_NONE;
            if (was_hit) finish_leaf<VOL>(p, best, ray, h);
            V3 term;
            const bool done = shade_segment(p, was_hit, h, ray, rng, lds_mats, chain, depth, term);
            if (done) { // the sample's clamped radiance (render.rs:95-97); k_sum_samples adds a pixel's samples in order
                const V3 rad = sample_radiance(p, lds_mats, chain, term);
                reinterpret_cast<float4 *>(p.samples)[st.w] = make_float4(rad.x, rad.y, rad.z, 0.0f);
                p.wf_stage[id] = 255u;
            } else {
                float4 *wq = reinterpret_cast<float4 *>(p.wf_ray) + 2ull * id;
                wq[0] = make_float4(ray.o.x, ray.o.y, ray.o.z, ray.d.x);
                wq[1] = make_float4(ray.d.y, ray.d.z, 0.0f, 0.0f);
                *reinterpret_cast<uint4 *>(p.wf_state + 4ull * id) = make_uint4(rng.state, rng.inc, chain.n, st.w);
                p.wf_stage[id] = (uint8_t)(depth0 + 1u);
            }
        }
        sc.tail += take;
    }
}

// Adds each pixel's samples in index order (render.rs:99) and finishes the pixel (render.rs:103-110), or stores the sum.
// A launch's samples may arrive in several batches (the buffer is bounded): `acc_io` carries the running sums from one
// batch to the next, so the additions stay those of one loop over all samples.
// A record whose fourth word has its top bit set is an escaping path's direction plus its packed attenuation chain
// (k_render_bvh_sp): SkyMap::sample (backgrounds.rs:84-92), the chain product (render.rs:52) and the clamp (render.rs:95-97)
// are evaluated here, by full waves.
__global__ __launch_bounds__(256) void k_sum_samples(const RenderParams p, const float *samples, uint32_t n_samples, float *acc_io,
                                                     uint32_t first_batch, uint32_t last_batch, uint32_t spp, uint32_t n_pixels,
                                                     float *sum_rgba, uint8_t *rgba8, float *mean_rgb) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_pixels) return;
    V3 acc = mk(0.0f, 0.0f, 0.0f);
    if (!first_batch) {
        const float4 c = reinterpret_cast<const float4 *>(acc_io)[i];
        acc = mk(c.x, c.y, c.z);
    }
    const float4 *src = reinterpret_cast<const float4 *>(samples) + i;
    if (p.bg_kind != 2u) { // no deferred records without a SkyMap
        for (uint32_t k = 0; k < n_samples; k++) {
            const float4 v = src[(uint64_t)k * n_pixels];
            acc = add(acc, mk(v.x, v.y, v.z));
        }
    } else {
        // four samples at a time: their loads, then their texel gathers, are issued together; the additions stay in index order
        for (uint32_t k = 0; k < n_samples; k += 4u) {
            float4 v[4], tx[4];
#pragma unroll
            for (uint32_t j = 0; j < 4u; j++)
                v[j] = k + j < n_samples ? src[(uint64_t)(k + j) * n_pixels] : make_float4(0.0f, 0.0f, 0.0f, 0.0f);
#pragma unroll
            for (uint32_t j = 0; j < 4u; j++) {
                const bool deferred = (__float_as_uint(v[j].w) & 0x80000000u) != 0u;
                const unsigned long long ti = deferred ? sky_texel_index(p, mk(v[j].x, v[j].y, v[j].z)) : 0ull;
                tx[j] = reinterpret_cast<const float4 *>(p.sky)[ti];
            }
#pragma unroll
            for (uint32_t j = 0; j < 4u; j++) {
                if (k + j >= n_samples) break;
                V3 rad = mk(v[j].x, v[j].y, v[j].z);
                const uint32_t w = __float_as_uint(v[j].w);
                if (w & 0x80000000u) {
                    V3 term = mk(tx[j].x, tx[j].y, tx[j].z);
                    for (uint32_t c = (w >> 24) & 3u; c > 0u; c--) // innermost attenuation first: the chain's last entry
                        term = mul(material_rgb(p, nullptr, (w >> (8u * (c - 1u))) & 255u), term);
                    rad = clamp_radiance(p, term);
                }
                acc = add(acc, rad);
            }
        }
    }
    if (!last_batch) {
        reinterpret_cast<float4 *>(acc_io)[i] = make_float4(acc.x, acc.y, acc.z, 0.0f);
        return;
    }
    store_pixel(acc, spp, i, sum_rgba, rgba8, mean_rgb);
}

// render.rs:103-110 for the sharded path: sum -> mean -> sqrt -> RGBA8
__global__ __launch_bounds__(256) void k_resolve(const float *sum_rgba, uint32_t spp, uint32_t n_pixels, uint8_t *rgba8,
                                                 float *mean_rgb) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_pixels) return;
    store_pixel(mk(sum_rgba[3ull * i], sum_rgba[3ull * i + 1], sum_rgba[3ull * i + 2]), spp, i, nullptr, rgba8, mean_rgb);
}

// rtr_create_multi with every rank on ONE device (a test configuration: RCCL cannot put two ranks on one GPU): the ranks'
// partial sums (three floats per pixel) added in rank order -- what ncclReduce does across GPUs.
__global__ __launch_bounds__(256) void k_sum_partials(const float *const *partials, uint32_t n, uint32_t n_floats, float *total) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_floats) return;
    float acc = partials[0][i];
    for (uint32_t r = 1; r < n; r++) acc += partials[r][i];
    total[i] = acc;
}
hipError_t launch_sum_partials(const float *const *d_partials, uint32_t n, uint32_t n_pixels, float *d_total, hipStream_t stream) {
    const uint32_t n_floats = n_pixels * 3u;
    hipLaunchKernelGGL(k_sum_partials, dim3((n_floats + 255u) / 256u), dim3(256), 0, stream, d_partials, n, n_floats, d_total);
    return hipGetLastError();
}

// rtr_probe_hits: world.hit (render.rs:38: tmin = 0.00001, tmax = +inf) for caller-supplied rays, one lane per ray, through
// the same world_hit the render kernels inline.  Closed-form known answers of Sphere::hit / Triangle::hit / the tree walk
// are asserted on the HIP path through it (tests/kat_cases.py), without the oracle in between.  A Volume's random draw
// (volume.rs:57) comes from the stream of (seed keys, pixel = ray index, sample 0).
// out: 12 words per ray = rtr_hit {hit, t, point[3], normal[3], front_face, material, 0, 0}.  LDS: the traversal stack.
template <uint32_t WORLD, uint32_t VOL, bool MESH>
__global__ __launch_bounds__(64) void k_probe_hits(const RenderParams p, const float *origins, const float *dirs, uint32_t n, uint32_t *out) {
    extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
    const uint32_t lane = threadIdx.x, i = blockIdx.x * 64u + lane;
    if (i >= n) return;
    uint32_t *stk_ref = lds;
    StackT *stk_t = reinterpret_cast<StackT *>(stk_ref + p.stack_cap * 64u);
    Ray ray;
    ray.o = ld3(origins + 3ull * i);
    ray.d = ld3(dirs + 3ull * i);
    const uint32_t hp = hash32(i ^ p.seed_key0);
    Rng rng;
    rng.inc = (hash32(hp + p.seed_key1) << 1) | 1u;
    rng.state = hash32(hp);
    Hit h;
    h.point = ray.o; h.normal = ray.o; h.t = 0.0f; h.front_face = false; h.material = 0u; h.slot = 0u;
    const SphereRegs<0> sr = {};
    const bool hit = world_hit<WORLD, 0u, VOL, MESH>(p, ray, h, stk_ref, stk_t, lane, sr, nullptr, rng);
    uint32_t *o = out + 12ull * i;
    o[0] = hit ? 1u : 0u;
    o[1] = hit ? __float_as_uint(h.t) : 0u;
    o[2] = __float_as_uint(h.point.x); o[3] = __float_as_uint(h.point.y); o[4] = __float_as_uint(h.point.z);
    o[5] = __float_as_uint(h.normal.x); o[6] = __float_as_uint(h.normal.y); o[7] = __float_as_uint(h.normal.z);
    o[8] = h.front_face ? 1u : 0u;
    o[9] = h.material;
    o[10] = 0u; o[11] = 0u;
}
hipError_t launch_probe_hits(const RenderParams &p, const float *d_origins, const float *d_dirs, uint32_t n, uint32_t *d_out, hipStream_t stream) {
    const uint32_t lds_bytes = p.stack_cap * 64u * (4u + (uint32_t)sizeof(StackT));
    const dim3 grid((n + 63u) / 64u), block(64);
    if (p.world == WORLD_BVH) hipLaunchKernelGGL((k_probe_hits<WORLD_BVH, 2u, true>), grid, block, lds_bytes, stream, p, d_origins, d_dirs, n, d_out);
    else if (p.world == WORLD_LINEAR_GENERIC) hipLaunchKernelGGL((k_probe_hits<WORLD_LINEAR_GENERIC, 2u, true>), grid, block, lds_bytes, stream, p, d_origins, d_dirs, n, d_out);
    else hipLaunchKernelGGL((k_probe_hits<WORLD_LINEAR_SPHERES, 0u, false>), grid, block, lds_bytes, stream, p, d_origins, d_dirs, n, d_out);
    return hipGetLastError();
}

// Exhaustive device self-test: rt_sqrt(x) against sqrtf(x) for every f32 bit pattern (NaNs compare as NaNs).
__global__ __launch_bounds__(256) void k_selftest_sqrt(unsigned long long *mismatches) {
    const uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x; // 2^20 threads x 4096 patterns
    uint32_t bad = 0;
    for (uint32_t i = 0; i < 4096u; i++) {
        const float x = __uint_as_float(tid * 4096u + i);
        const float a = rt_sqrt(x), b = sqrtf(x);
        const bool same = (__float_as_uint(a) == __float_as_uint(b)) || (a != a && b != b);
        bad += same ? 0u : 1u;
    }
    if (bad) atomicAdd(mismatches, (unsigned long long)bad);
}

// Device self-test: shared_div(n, d) against n / d for every f32 bit pattern n in [lo_bits, hi_bits].
__global__ __launch_bounds__(256) void k_selftest_div(float d, uint32_t lo_bits, uint32_t hi_bits, unsigned long long *mismatches) {
    const SharedDiv sd = shared_div_prepare(d);
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    uint32_t bad = sd.ok ? 0u : 1u;
    for (uint64_t b = (uint64_t)lo_bits + blockIdx.x * blockDim.x + threadIdx.x; b <= hi_bits; b += stride) {
        const float n = __uint_as_float((uint32_t)b);
        const float q = shared_div(n, sd.d, sd.r1), ref = n / d;
        bad += (__float_as_uint(q) == __float_as_uint(ref)) ? 0u : 1u;
    }
    if (bad) atomicAdd(mismatches, (unsigned long long)bad);
}

// option "debug_fault": sets bit 1 of the context's fault word, as a chunk that timed out sets bit 0 (tests the
// path from the device word to the C ABI's error code without a kernel that misbehaves)
__global__ void k_debug_fault(unsigned long long *fault) { atomicOr(fault, 2ull); }
hipError_t launch_debug_fault(unsigned long long *fault, hipStream_t stream) {
    hipLaunchKernelGGL(k_debug_fault, dim3(1), dim3(1), 0, stream, fault);
    return hipGetLastError();
}

hipError_t launch_selftest_div(float d, uint32_t lo_bits, uint32_t hi_bits, unsigned long long *d_mismatches, hipStream_t stream) {
    hipLaunchKernelGGL(k_selftest_div, dim3(8192), dim3(256), 0, stream, d, lo_bits, hi_bits, d_mismatches);
    return hipGetLastError();
}

hipError_t launch_selftest_sqrt(unsigned long long *d_mismatches, hipStream_t stream) {
    hipLaunchKernelGGL(k_selftest_sqrt, dim3(4096), dim3(256), 0, stream, d_mismatches);
    return hipGetLastError();
}

// --- launchers (called from rtr_api.cpp) -----------------------------------------------------------------
// One table of kernel entry points; launching, the occupancy queries and the attribute queries all go through it.
// Tree kernels are instantiated per Volume level (0: no Volume objects; 1: sphere-bounded ones -- the reference's visit
// set and order, the random draw inside world.hit; 2: mesh-bounded ones too -- adds the nested mesh walk) and per
// "scene has triangles" (unconditional slab tests, see RTR_WALK_TRIP).
#define RTR_FN(...) reinterpret_cast<const void *>(&__VA_ARGS__)
#define RTR_TREE_TABLE(KERNEL)                                                                                       \
    {{RTR_FN(KERNEL<0u, false>), RTR_FN(KERNEL<0u, true>)}, {RTR_FN(KERNEL<1u, false>), RTR_FN(KERNEL<1u, true>)},  \
     {RTR_FN(KERNEL<2u, false>), RTR_FN(KERNEL<2u, true>)}}
static uint32_t vol_level(const RenderParams &p) { return p.has_volumes > 2u ? 2u : p.has_volumes; }

static const void *render_kernel(const RenderParams &p) {
    const uint32_t v = vol_level(p), m = p.has_meshes ? 1u : 0u;
    if (p.world == WORLD_BVH && p.sample_parallel) {
        static const void *const wide[3][2] = {{RTR_FN(k_render_bvh_sp<0u, false, false>), RTR_FN(k_render_bvh_sp<0u, true, false>)},
                                               {RTR_FN(k_render_bvh_sp<1u, false, false>), RTR_FN(k_render_bvh_sp<1u, true, false>)},
                                               {RTR_FN(k_render_bvh_sp<2u, false, false>), RTR_FN(k_render_bvh_sp<2u, true, false>)}};
        static const void *const packed[3][2] = {{RTR_FN(k_render_bvh_sp<0u, false, true>), RTR_FN(k_render_bvh_sp<0u, true, true>)},
                                                 {RTR_FN(k_render_bvh_sp<1u, false, true>), RTR_FN(k_render_bvh_sp<1u, true, true>)},
                                                 {RTR_FN(k_render_bvh_sp<2u, false, true>), RTR_FN(k_render_bvh_sp<2u, true, true>)}};
        if (p.walk_counters) { // option "traversal_counters": the counting instantiations
            static const void *const cw[3][2] = {{RTR_FN(k_render_bvh_sp<0u, false, false, true>), RTR_FN(k_render_bvh_sp<0u, true, false, true>)},
                                                 {RTR_FN(k_render_bvh_sp<1u, false, false, true>), RTR_FN(k_render_bvh_sp<1u, true, false, true>)},
                                                 {RTR_FN(k_render_bvh_sp<2u, false, false, true>), RTR_FN(k_render_bvh_sp<2u, true, false, true>)}};
            static const void *const cp[3][2] = {{RTR_FN(k_render_bvh_sp<0u, false, true, true>), RTR_FN(k_render_bvh_sp<0u, true, true, true>)},
                                                 {RTR_FN(k_render_bvh_sp<1u, false, true, true>), RTR_FN(k_render_bvh_sp<1u, true, true, true>)},
                                                 {RTR_FN(k_render_bvh_sp<2u, false, true, true>), RTR_FN(k_render_bvh_sp<2u, true, true, true>)}};
            return p.stack_packed ? cp[v][m] : cw[v][m];
        }
        return p.stack_packed ? packed[v][m] : wide[v][m];
    }
    if (p.world == WORLD_BVH && p.resumable_walk) {
        static const void *const t[3][2] = RTR_TREE_TABLE(k_render_bvh);
        return t[v][m];
    }
    if (p.world == WORLD_BVH) {
        static const void *const t[3][2] = {
            {RTR_FN(k_render<WORLD_BVH, 0, 0u, false>), RTR_FN(k_render<WORLD_BVH, 0, 0u, true>)},
            {RTR_FN(k_render<WORLD_BVH, 0, 1u, false>), RTR_FN(k_render<WORLD_BVH, 0, 1u, true>)},
            {RTR_FN(k_render<WORLD_BVH, 0, 2u, false>), RTR_FN(k_render<WORLD_BVH, 0, 2u, true>)}};
        return t[v][m];
    }
    if (p.world == WORLD_LINEAR_GENERIC) {
        static const void *const t[3] = {RTR_FN(k_render<WORLD_LINEAR_GENERIC, 0, 0u, true>), RTR_FN(k_render<WORLD_LINEAR_GENERIC, 0, 1u, true>),
                                         RTR_FN(k_render<WORLD_LINEAR_GENERIC, 0, 2u, true>)};
        return t[v];
    }
    if (p.plain_sphere_world) { // Full mode, uniform background, no Mix / Isotropic materials, attenuation chain: shade_segment<PLAIN>
        static const void *const q[9] = {
            nullptr, RTR_FN(k_render<WORLD_LINEAR_SPHERES, 1, 0u, false, true>),
            RTR_FN(k_render<WORLD_LINEAR_SPHERES, 2, 0u, false, true>), RTR_FN(k_render<WORLD_LINEAR_SPHERES, 3, 0u, false, true>),
            RTR_FN(k_render<WORLD_LINEAR_SPHERES, 4, 0u, false, true>), RTR_FN(k_render<WORLD_LINEAR_SPHERES, 5, 0u, false, true>),
            RTR_FN(k_render<WORLD_LINEAR_SPHERES, 6, 0u, false, true>), RTR_FN(k_render<WORLD_LINEAR_SPHERES, 7, 0u, false, true>),
            RTR_FN(k_render<WORLD_LINEAR_SPHERES, 8, 0u, false, true>)};
        return q[p.sphere_regs <= 8u ? p.sphere_regs : 0u];
    }
    static const void *const t[9] = { // number of spheres held in scalar registers (0 = generic loop)
        RTR_FN(k_render<WORLD_LINEAR_SPHERES, 0, 0u, false>), RTR_FN(k_render<WORLD_LINEAR_SPHERES, 1, 0u, false>),
        RTR_FN(k_render<WORLD_LINEAR_SPHERES, 2, 0u, false>), RTR_FN(k_render<WORLD_LINEAR_SPHERES, 3, 0u, false>),
        RTR_FN(k_render<WORLD_LINEAR_SPHERES, 4, 0u, false>), RTR_FN(k_render<WORLD_LINEAR_SPHERES, 5, 0u, false>),
        RTR_FN(k_render<WORLD_LINEAR_SPHERES, 6, 0u, false>), RTR_FN(k_render<WORLD_LINEAR_SPHERES, 7, 0u, false>),
        RTR_FN(k_render<WORLD_LINEAR_SPHERES, 8, 0u, false>)};
    return t[p.sphere_regs <= 8u ? p.sphere_regs : 0u];
}
static const void *wf_intersect_kernel(const RenderParams &p) {
    static const void *const t[3][2] = RTR_TREE_TABLE(k_wf_intersect);
    return t[vol_level(p)][p.has_meshes ? 1u : 0u];
}
static const void *wf_shade_kernel(const RenderParams &p) {
    static const void *const t[3] = {RTR_FN(k_wf_shade<0u>), RTR_FN(k_wf_shade<1u>), RTR_FN(k_wf_shade<2u>)};
    return t[vol_level(p)];
}
static hipError_t launch_params_kernel(const void *fn, const RenderParams &p, dim3 grid, uint32_t lds_bytes, hipStream_t stream) {
    RenderParams copy = p;
    void *args[] = {&copy};
    return hipLaunchKernel(fn, grid, dim3(64), args, lds_bytes, stream);
}

hipError_t launch_render(const RenderParams &p, uint32_t grid, uint32_t lds_bytes, hipStream_t stream) {
    const void *fn = render_kernel(p);
    if (!fn) return hipErrorInvalidDeviceFunction;
    return launch_params_kernel(fn, p, dim3(grid), lds_bytes, stream);
}

hipError_t launch_wf_generate(const RenderParams &p, uint32_t batch_samples, hipStream_t stream) {
    hipLaunchKernelGGL(k_wf_generate, dim3(p.tiles_total, batch_samples), dim3(64), 0, stream, p);
    return hipGetLastError();
}
uint32_t stack_entry_bytes(const RenderParams &p) { return p.stack_packed ? 4u : 4u + (uint32_t)sizeof(StackT); }
uint32_t wf_intersect_lds_bytes(const RenderParams &p) { return p.stack_cap * 64u * stack_entry_bytes(p) + kWfRing * 4u; }
uint32_t wf_shade_lds_bytes(const RenderParams &p) { return (kWfRing + p.lds_materials * 8u) * 4u; }
static uint32_t resident_waves(const void *fn, uint32_t lds_bytes, int block = 64) { // resident WORKGROUPS of `block` threads
    int per_cu = 0, dev = 0, cus = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, fn, block, lds_bytes) != hipSuccess || per_cu <= 0) per_cu = 1024 / block;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) cus = 256;
    return (uint32_t)per_cu * (uint32_t)cus;
}
// workgroups (one wave each) of k_wf_intersect / k_wf_shade the device keeps resident
uint32_t wf_resident_waves(const RenderParams &p, bool shade) {
    return shade ? resident_waves(wf_shade_kernel(p), wf_shade_lds_bytes(p)) : resident_waves(wf_intersect_kernel(p), wf_intersect_lds_bytes(p));
}
hipError_t launch_wf_intersect(const RenderParams &p, uint32_t grid, hipStream_t stream) {
    return launch_params_kernel(wf_intersect_kernel(p), p, dim3(grid), wf_intersect_lds_bytes(p), stream);
}
hipError_t launch_wf_shade(const RenderParams &p, uint32_t grid, hipStream_t stream) {
    return launch_params_kernel(wf_shade_kernel(p), p, dim3(grid), wf_shade_lds_bytes(p), stream);
}
hipError_t wf_intersect_attributes(const RenderParams &p, int *vgprs) {
    hipFuncAttributes attr;
    hipError_t e = hipFuncGetAttributes(&attr, wf_intersect_kernel(p));
    if (e == hipSuccess) *vgprs = attr.numRegs;
    return e;
}

hipError_t launch_sum_samples(const RenderParams &p, const float *samples, uint32_t n_samples, float *acc_io, bool first_batch, bool last_batch, uint32_t spp,
                              uint32_t n_pixels, float *sum_rgba, uint8_t *rgba8, float *mean_rgb, hipStream_t stream) {
    hipLaunchKernelGGL(k_sum_samples, dim3((n_pixels + 255u) / 256u), dim3(256), 0, stream, p, samples, n_samples, acc_io,
                       first_batch ? 1u : 0u, last_batch ? 1u : 0u, spp, n_pixels, sum_rgba, rgba8, mean_rgb);
    return hipGetLastError();
}

hipError_t launch_resolve(const float *sum_rgba, uint32_t spp, uint32_t n_pixels, uint8_t *rgba8, float *mean_rgb,
                          hipStream_t stream) {
    uint32_t grid = (n_pixels + 255u) / 256u;
    hipLaunchKernelGGL(k_resolve, dim3(grid), dim3(256), 0, stream, sum_rgba, spp, n_pixels, rgba8, mean_rgb);
    return hipGetLastError();
}

// 64-thread workgroups of the kernel launch_render would take that the current device keeps resident
uint32_t render_resident_waves(const RenderParams &p, uint32_t lds_bytes) { return resident_waves(render_kernel(p), lds_bytes); }

hipError_t render_kernel_attributes(const RenderParams &p, int *vgprs, int *static_lds) {
    hipFuncAttributes attr;
    const void *fn = render_kernel(p);
    hipError_t e = hipFuncGetAttributes(&attr, fn);
    if (e != hipSuccess) return e;
    *vgprs = attr.numRegs;
    *static_lds = (int)attr.sharedSizeBytes;
    return hipSuccess;
}

// the name rocprofv3 prints for the kernel launch_render would take (bench.py's roofline.kernel)
const char *render_kernel_name(const RenderParams &p) {
    if (p.world == WORLD_BVH && p.wf_ray) return "rtr::k_wf_intersect (tree world, streaming wavefront: paths in HBM) + rtr::k_wf_generate / k_wf_shade / k_sum_samples";
    if (p.world == WORLD_BVH && p.sample_parallel) return "rtr::k_render_bvh_sp (tree world, lanes take samples) + rtr::k_sum_samples";
    if (p.world == WORLD_BVH && p.resumable_walk) return "rtr::k_render_bvh (tree world, resumable walk)";
    if (p.world == WORLD_BVH) return "rtr::k_render<WORLD_BVH> (tree world, wave-synchronous walk)";
    if (p.world == WORLD_LINEAR_GENERIC) return "rtr::k_render<WORLD_LINEAR_GENERIC> (linear world with meshes)";
    return p.sphere_regs ? "rtr::k_render<WORLD_LINEAR_SPHERES, NS> (linear world, spheres in scalar registers)"
                         : "rtr::k_render<WORLD_LINEAR_SPHERES, 0> (linear world, generic sphere loop)";
}

} // namespace rtr
