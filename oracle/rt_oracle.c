/*
 * rt_oracle.c -- CPU restatement of LevKruglyak/RayTracingRust's per-pixel radiance integrator.
 *
 * TEST INFRASTRUCTURE ONLY (see rt_oracle.h): the parity checker for the HIP path and the "port"
 * CPU baseline.  "parity unpinned": the reference has no golden vectors and cannot be built here;
 * this file is pinned by RNG-free known-answer tests only (tests/test_oracle_kat.py).
 *
 * Every function cites the reference file:line it restates (paths relative to /root/reference).
 * The data structures deliberately mirror the reference's (boxed pointer tree, recursive unculled
 * traversal, recursive trace_ray) so that operation order -- and therefore every f32 rounding --
 * is the reference's.  Build with -ffp-contract=off (oracle/Makefile): Rust never contracts a*b+c.
 *
 * Third-party arithmetic restated from published behaviour (crates absent from the tree, no
 * Cargo.lock; versions from Cargo.toml:42-61):
 *   cgmath 0.18  Vector3<f32>: dot = (x*x'+y*y')+z*z'; cross; magnitude = sqrt(dot(v,v));
 *                normalize = v * (1/|v|); v/s per-component divide; angle = atan2(|a x b|, a.b)
 *   rand 0.8     floats from the top 23 bits of a u32: Uniform(0..=1) = k*2^-23 * (1+2^-23),
 *                gen_range(lo..hi) = k*2^-23*(hi-lo)+lo.  The generator itself is NOT rand's
 *                (thread_rng is OS-seeded ChaCha12): see or_rng_* below.
 */
#define _GNU_SOURCE
#include "rt_oracle.h"

#include <math.h>
#include <pthread.h>
#include <stdatomic.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include <unistd.h>

/* ------------------------------------------------------------------------------------------ */
/* lib/utils/types.rs:4-8  Float = f32, PI                                                      */
typedef struct { float x, y, z; } v3;
#define OR_PI 3.14159265358979323846f

static __thread char g_err[256];
const char *or_last_error(void) { return g_err; }
static int or_fail(int code, const char *msg) {
    snprintf(g_err, sizeof g_err, "%s", msg);
    return code;
}

/* cgmath 0.18 Vector3<f32> operators */
static inline v3 v3_new(float x, float y, float z) { v3 r = {x, y, z}; return r; }
static inline v3 v3_add(v3 a, v3 b) { return v3_new(a.x + b.x, a.y + b.y, a.z + b.z); }
static inline v3 v3_sub(v3 a, v3 b) { return v3_new(a.x - b.x, a.y - b.y, a.z - b.z); }
static inline v3 v3_scale(v3 a, float s) { return v3_new(a.x * s, a.y * s, a.z * s); }
static inline v3 v3_divs(v3 a, float s) { return v3_new(a.x / s, a.y / s, a.z / s); }
static inline v3 v3_neg(v3 a) { return v3_new(-a.x, -a.y, -a.z); }
static inline float v3_dot(v3 a, v3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }
static inline v3 v3_cross(v3 a, v3 b) {
    return v3_new((a.y * b.z) - (a.z * b.y), (a.z * b.x) - (a.x * b.z), (a.x * b.y) - (a.y * b.x));
}
static inline float v3_mag2(v3 a) { return v3_dot(a, a); }
static inline float v3_mag(v3 a) { return sqrtf(v3_dot(a, a)); }
static inline v3 v3_normalize(v3 a) { return v3_scale(a, 1.0f / v3_mag(a)); }

/* lib/utils/color.rs:5-81 */
typedef struct { float r, g, b; } col;
static inline col col_new(float r, float g, float b) { col c = {r, g, b}; return c; }
static inline col col_add(col a, col b) { return col_new(a.r + b.r, a.g + b.g, a.b + b.b); }
static inline col col_mul(col a, col b) { return col_new(a.r * b.r, a.g * b.g, a.b * b.b); }
static inline col col_scale(col a, float s) { return col_new(a.r * s, a.g * s, a.b * s); }

/* Rust `as u8` on f32: truncate toward zero, saturate, NaN -> 0 (color.rs:34-41) */
static inline uint8_t sat_u8(float f) {
    if (!(f == f)) return 0;
    if (f <= 0.0f) return 0;
    if (f >= 255.0f) return 255;
    return (uint8_t)f;
}
/* Rust `as usize` on f32 (backgrounds.rs:88-89) */
static inline uint64_t sat_usize(float f) {
    if (!(f == f)) return 0;
    if (f <= 0.0f) return 0;
    if (f >= 18446744073709551616.0f) return UINT64_MAX;
    return (uint64_t)f;
}

/* lib/utils/ray.rs:8-17 */
typedef struct { v3 origin, direction; } ray_t;
static inline v3 ray_at(const ray_t *r, float t) { return v3_add(r->origin, v3_scale(r->direction, t)); }

/* lib/utils/ray.rs:29-55  HitRecord::new */
typedef struct { v3 point, normal; float t; int front_face; int material; } hit_t;
static inline hit_t hit_new(v3 point, v3 outward_normal, float t, const ray_t *ray, int material) {
    hit_t h;
    h.front_face = v3_dot(ray->direction, outward_normal) < 0.0f;
    h.normal = h.front_face ? outward_normal : v3_neg(outward_normal);
    h.point = point;
    h.t = t;
    h.material = material;
    return h;
}

/* ------------------------------------------------------------------------------------------ */
/* Random numbers.  The reference draws from rand::thread_rng() everywhere (render.rs:80,
 * materials.rs:127,151, sample.rs:60,78, volume.rs:57): OS-seeded, so nothing can pin its values.
 * We substitute one counter-keyed stream per (seed, pixel, sample): a 32-bit PCG (RXS-M-XS output,
 * per-pixel odd increment); both halves of the seed are hashed into both stream keys, so frames of
 * different seeds are never pixel permutations of each other.  The rejection sampler's candidates
 * (sample.rs:43-51) come from a sub-stream: one word of the path's stream seeds a plain 32-bit LCG
 * whose successive states are the coordinates (or_ball_coord).  The float conversions are rand 0.8's
 * (top 23 bits).                                                                              */
typedef struct { uint32_t state, inc; uint64_t draws; } or_rng;

static inline uint32_t or_hash32(uint32_t x) {
    x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16;
    return x;
}
static inline void or_rng_seed(or_rng *r, uint64_t seed, uint32_t pixel, uint32_t sample) {
    uint32_t lo = (uint32_t)seed, hi = (uint32_t)(seed >> 32);
    uint32_t k0 = or_hash32(lo ^ or_hash32(hi + 0x9E3779B9u)), k1 = or_hash32(hi ^ or_hash32(lo + 0x85EBCA6Bu));
    uint32_t hp = or_hash32(pixel ^ k0);
    r->inc = (or_hash32(hp + k1) << 1) | 1u;
    r->state = or_hash32(hp + sample * 0x9E3779B9u);
}
static inline uint32_t or_rng_u32(or_rng *r) {
    uint32_t old = r->state;
    r->state = old * 747796405u + r->inc;
    uint32_t word = ((old >> ((old >> 28) + 4u)) ^ old) * 277803737u;
    r->draws++;
    return (word >> 22) ^ word;
}
/* value0_1 = k * 2^-23, k = top 23 bits */
static inline float or_rng_value01(or_rng *r) { return (float)(or_rng_u32(r) >> 9) * 0x1p-23f; }
/* Uniform::from(0.0..=1.0).sample (render.rs:81,88-89): scale = 1/(1-2^-23) -> 1+2^-23 in f32 */
static inline float or_rng_unit_inclusive(or_rng *r) { return or_rng_value01(r) * 0x1.000002p+0f + 0.0f; }
/* gen_range(0.0..1.0) (materials.rs:127,151) */
static inline float or_rng_unit(or_rng *r) { return or_rng_value01(r) * 1.0f + 0.0f; }
/* gen_range(-1.0..1.0) (sample.rs:29-31 via CubeSampler::from_range(-1.0..=1.0)) on the next state of the
 * rejection sampler's LCG sub-stream: value0_1 * 2 + -1 with value0_1 = (top 23 bits) * 2^-23 */
static inline float or_ball_coord(uint32_t *s) {
    *s = *s * 747796405u + 2891336453u;
    return ((float)(*s >> 9) * 0x1p-23f) * 2.0f + -1.0f;
}

/* ------------------------------------------------------------------------------------------ */
/* lib/utils/aabb.rs */
typedef struct { v3 min, max; } aabb_t;
static inline aabb_t aabb_default(void) { aabb_t b = {{0, 0, 0}, {0, 0, 0}}; return b; } /* aabb.rs:140-147 */
static inline aabb_t aabb_from_point(v3 p) { aabb_t b = {p, p}; return b; }             /* aabb.rs:21-26 */
static inline v3 aabb_centroid(aabb_t b) { return v3_divs(v3_add(b.min, b.max), 2.0f); }  /* aabb.rs:28-30 */
static inline aabb_t aabb_surround(aabb_t a, aabb_t b) {                                  /* aabb.rs:41-54 */
    aabb_t r;
    r.min = v3_new(fminf(a.min.x, b.min.x), fminf(a.min.y, b.min.y), fminf(a.min.z, b.min.z));
    r.max = v3_new(fmaxf(a.max.x, b.max.x), fmaxf(a.max.y, b.max.y), fmaxf(a.max.z, b.max.z));
    return r;
}
static aabb_t aabb_epsilon_expand(aabb_t bounds, float epsilon) { /* aabb.rs:57-78 */
    v3 dim = v3_new(bounds.max.x - bounds.min.x, bounds.max.y - bounds.min.y, bounds.max.z - bounds.min.z);
    v3 c = aabb_centroid(bounds);
    aabb_t r = bounds;
    if (dim.x < epsilon) { r.min.x = c.x - epsilon; r.max.x = c.x + epsilon; }
    if (dim.y < epsilon) { r.min.y = c.y - epsilon; r.max.y = c.y + epsilon; }
    if (dim.z < epsilon) { r.min.z = c.z - epsilon; r.max.z = c.z + epsilon; }
    return r;
}
/* aabb.rs:81-137: three sequential slab passes, early return after each */
static int aabb_hit(const aabb_t *b, const ray_t *ray, float tmin, float tmax) {
    float inv_d, t0, t1, tmp;
    inv_d = 1.0f / ray->direction.x;
    t0 = (b->min.x - ray->origin.x) * inv_d;
    t1 = (b->max.x - ray->origin.x) * inv_d;
    if (inv_d < 0.0f) { tmp = t0; t0 = t1; t1 = tmp; }
    if (t0 > tmin) tmin = t0;
    if (t1 < tmax) tmax = t1;
    if (tmax <= tmin) return 0;

    inv_d = 1.0f / ray->direction.y;
    t0 = (b->min.y - ray->origin.y) * inv_d;
    t1 = (b->max.y - ray->origin.y) * inv_d;
    if (inv_d < 0.0f) { tmp = t0; t0 = t1; t1 = tmp; }
    if (t0 > tmin) tmin = t0;
    if (t1 < tmax) tmax = t1;
    if (tmax <= tmin) return 0;

    inv_d = 1.0f / ray->direction.z;
    t0 = (b->min.z - ray->origin.z) * inv_d;
    t1 = (b->max.z - ray->origin.z) * inv_d;
    if (inv_d < 0.0f) { tmp = t0; t0 = t1; t1 = tmp; }
    if (t0 > tmin) tmin = t0;
    if (t1 < tmax) tmax = t1;
    if (tmax <= tmin) return 0;
    return 1;
}

/* ------------------------------------------------------------------------------------------ */
/* scene data                                                                                   */
typedef struct bvh_node { /* bvh.rs:52-56 */
    int kind;             /* 0 None, 1 Object(handle), 2 Split(bounds,left,right) */
    uint32_t handle;
    aabb_t bounds;
    struct bvh_node *left, *right;
} bvh_node;

typedef struct { v3 center; float radius; int material; } sphere_t; /* objects.rs:15-20 */
typedef struct { v3 position, normal; } vertex_t;                   /* mesh.rs:25-29 */
typedef struct { uint32_t v[3]; v3 normal; } triangle_t;            /* mesh.rs:19-23 */
typedef struct {                                                    /* mesh.rs:31-38 */
    vertex_t *vertices; uint32_t n_verts;
    aabb_t bounds;
    triangle_t *triangles; uint32_t n_tris;
    bvh_node *bvh_root;
    int material;
} mesh_t;
/* volume.rs:19-23: `boundary: Box<dyn Object>` -- a sphere, a mesh (boundary_mesh != NULL; the mesh's material is the
 * medium's, Object::material of Mesh) or another Volume (boundary_volume != NULL, to any depth) */
typedef struct volume_s { sphere_t boundary; mesh_t *boundary_mesh; struct volume_s *boundary_volume; float neg_inv_density; } volume_t;
static int volume_material(const volume_t *v) { /* Object::material = boundary.material(), volume.rs:98-100 */
    return v->boundary_volume ? volume_material(v->boundary_volume) : (v->boundary_mesh ? v->boundary_mesh->material : v->boundary.material);
}

enum { OBJ_SPHERE = 0, OBJ_MESH = 1, OBJ_VOLUME = 2 };
typedef struct { int kind; sphere_t sphere; mesh_t *mesh; volume_t volume; } object_t;

typedef struct { int kind; col color; float param; int first, second; } material_t;

struct or_scene {
    /* camera.rs:27-34 */
    v3 lookfrom, lookat, vertical; float vertical_fov, aspect_ratio;
    /* scene.rs:22-30 */
    uint32_t samples_per_pixel; uint32_t max_ray_depth; float clamp_indirect;
    int enable_multithreading, enable_bvh_tree, mode;
    int trig_mode;
    int sky_bilinear; /* or_set_sky_filter: an opt-in that is NOT reference behaviour (SkyMap::sample takes the nearest texel) */
    /* background */
    int bg_kind; col bg_a, bg_b; col *sky; uint32_t sky_w, sky_h;
    object_t *objects; int n_objects, cap_objects;
    material_t *materials; int n_materials, cap_materials;
};

typedef struct { /* per-thread state */
    or_rng rng;
    uint64_t segments, box_tests, sphere_tests, tri_tests, sky_lookups, draws;
} tctx_t;

/* ------------------------------------------------------------------------------------------ */
/* lib/objects.rs:22-51  Sphere::hit */
static int sphere_hit(const sphere_t *s, const ray_t *ray, float tmin, float tmax, hit_t *out, tctx_t *tc) {
    tc->sphere_tests++;
    v3 oc = v3_sub(ray->origin, s->center);
    float a = v3_mag2(ray->direction);
    float half_b = v3_dot(oc, ray->direction);
    float c = v3_mag2(oc) - s->radius * s->radius;
    float discriminant = half_b * half_b - a * c;
    if (discriminant < 0.0f) return 0;
    float sqrtd = sqrtf(discriminant);
    float root = (-half_b - sqrtd) / a;
    if (root < tmin || tmax < root) {
        root = (-half_b + sqrtd) / a;
        if (root < tmin || tmax < root) return 0;
    }
    v3 point = ray_at(ray, root);
    v3 normal = v3_divs(v3_sub(point, s->center), s->radius);
    *out = hit_new(point, normal, root, ray, s->material);
    return 1;
}
/* objects.rs:53-60 */
static aabb_t sphere_bounds(const sphere_t *s) {
    v3 r = v3_new(s->radius, s->radius, s->radius);
    aabb_t b = {v3_sub(s->center, r), v3_add(s->center, r)};
    return b;
}

/* lib/utils/math.rs:31-33 */
static inline float degrees_to_radians(float d) { return d * OR_PI / 180.0f; }

/* lib/core/mesh.rs:146-197  Triangle::hit (tmax unused) */
static int triangle_hit(const triangle_t *tri, const mesh_t *mesh, const ray_t *ray, float tmin, hit_t *out,
                        tctx_t *tc) {
    tc->tri_tests++;
    const vertex_t *v0 = &mesh->vertices[tri->v[0]];
    const vertex_t *v1 = &mesh->vertices[tri->v[1]];
    const vertex_t *v2 = &mesh->vertices[tri->v[2]];
    v3 e1 = v3_sub(v1->position, v0->position);
    v3 e2 = v3_sub(v2->position, v0->position);
    v3 h = v3_cross(ray->direction, e2);
    float a = v3_dot(e1, h);
    if (a > -tmin && a < tmin) return 0;
    float f = 1.0f / a;
    v3 s = v3_sub(ray->origin, v0->position);
    float u = f * v3_dot(s, h);
    if (u < 0.0f || u > 1.0f) return 0;
    v3 q = v3_cross(s, e1);
    float v = f * v3_dot(ray->direction, q);
    if (v < 0.0f || u + v > 1.0f) return 0;
    float t = f * v3_dot(e2, q);
    v3 normal = v3_add(v3_add(v3_scale(v1->normal, u), v3_scale(v2->normal, v)), v3_scale(v0->normal, 1.0f - u - v));
    /* mesh.rs:187: cgmath Vector3::angle = atan2(|a x b|, a.b) */
    float angle = atan2f(v3_mag(v3_cross(normal, tri->normal)), v3_dot(normal, tri->normal));
    if (angle > degrees_to_radians(0.0f)) normal = tri->normal;
    if (t > tmin) {
        *out = hit_new(ray_at(ray, t), normal, t, ray, mesh->material);
        return 1;
    }
    return 0;
}
/* mesh.rs:201-213  BoundsCollection::bounds for Mesh */
static aabb_t mesh_tri_bounds(const mesh_t *m, uint32_t handle) {
    const triangle_t *t = &m->triangles[handle];
    return aabb_epsilon_expand(
        aabb_surround(aabb_from_point(m->vertices[t->v[0]].position),
                      aabb_surround(aabb_from_point(m->vertices[t->v[1]].position),
                                    aabb_from_point(m->vertices[t->v[2]].position))),
        0.01f);
}

/* ------------------------------------------------------------------------------------------ */
/* BoundsCollection (bvh.rs:18-28): Scene (scene.rs:117-139) and Mesh (mesh.rs:200-232)         */
typedef struct collection {
    const or_scene *scene; /* Scene collection */
    const mesh_t *mesh;    /* Mesh collection  */
} collection;

static int object_hit(const or_scene *sc, const object_t *o, const ray_t *ray, float tmin, float tmax, hit_t *out,
                      tctx_t *tc);

static aabb_t coll_bounds(const collection *c, uint32_t handle) {
    if (c->mesh) return mesh_tri_bounds(c->mesh, handle);
    const object_t *o = &c->scene->objects[handle];
    switch (o->kind) {
    case OBJ_SPHERE: return sphere_bounds(&o->sphere);
    case OBJ_MESH: return o->mesh->bounds;                        /* mesh.rs:234-238 */
    default: { /* Bounded for Volume = boundary.bounds(), volume.rs:90-94 */
        const volume_t *v = &o->volume;
        while (v->boundary_volume) v = v->boundary_volume;
        return v->boundary_mesh ? v->boundary_mesh->bounds : sphere_bounds(&v->boundary);
    }
    }
}
static int coll_hit(const collection *c, uint32_t handle, const ray_t *ray, float tmin, float tmax, hit_t *out,
                    tctx_t *tc) {
    if (c->mesh) return triangle_hit(&c->mesh->triangles[handle], c->mesh, ray, tmin, out, tc); /* mesh.rs:223-231 */
    return object_hit(c->scene, &c->scene->objects[handle], ray, tmin, tmax, out, tc);           /* scene.rs:130-138 */
}

/* bvh.rs:59-144  BvhNode::from_list.  *err set when partial_cmp().unwrap() would panic (NaN). */
static void bvh_free(bvh_node *n) {
    if (!n) return;
    bvh_free(n->left);
    bvh_free(n->right);
    free(n);
}
static void stable_sort_by_key(uint32_t *objs, float *keys, int n, uint32_t *tmp_o, float *tmp_k) {
    /* slice::sort_by is a stable merge sort; any stable sort yields the same permutation */
    if (n < 2) return;
    int mid = n / 2;
    stable_sort_by_key(objs, keys, mid, tmp_o, tmp_k);
    stable_sort_by_key(objs + mid, keys + mid, n - mid, tmp_o, tmp_k);
    int i = 0, j = mid, k = 0;
    while (i < mid && j < n) {
        if (keys[j] < keys[i]) { tmp_o[k] = objs[j]; tmp_k[k++] = keys[j++]; }   /* strict: equal keeps left first */
        else { tmp_o[k] = objs[i]; tmp_k[k++] = keys[i++]; }
    }
    while (i < mid) { tmp_o[k] = objs[i]; tmp_k[k++] = keys[i++]; }
    while (j < n) { tmp_o[k] = objs[j]; tmp_k[k++] = keys[j++]; }
    memcpy(objs, tmp_o, (size_t)n * sizeof *objs);
    memcpy(keys, tmp_k, (size_t)n * sizeof *keys);
}
static bvh_node *bvh_from_list(uint32_t *objects, int n, const collection *coll, int *err) {
    bvh_node *node = (bvh_node *)calloc(1, sizeof *node);
    if (n == 0) { node->kind = 0; return node; }
    if (n == 1) { node->kind = 1; node->handle = objects[0]; return node; }

    aabb_t bounds = coll_bounds(coll, objects[0]);                      /* reduce: seeded by the first element */
    for (int i = 1; i < n; i++) bounds = aabb_surround(bounds, coll_bounds(coll, objects[i]));
    aabb_t centroids = aabb_from_point(aabb_centroid(coll_bounds(coll, objects[0])));
    for (int i = 1; i < n; i++)
        centroids = aabb_surround(centroids, aabb_from_point(aabb_centroid(coll_bounds(coll, objects[i]))));

    v3 spread = v3_sub(centroids.max, centroids.min);
    int axis;
    if (spread.x > spread.y && spread.x > spread.z) axis = 0;
    else if (spread.y > spread.x && spread.y > spread.z) axis = 1;
    else axis = 2;

    float *keys = (float *)malloc((size_t)n * sizeof *keys);
    float *tmp_k = (float *)malloc((size_t)n * sizeof *tmp_k);
    uint32_t *tmp_o = (uint32_t *)malloc((size_t)n * sizeof *tmp_o);
    for (int i = 0; i < n; i++) {
        v3 c = aabb_centroid(coll_bounds(coll, objects[i]));
        keys[i] = axis == 0 ? c.x : (axis == 1 ? c.y : c.z);
        if (!(keys[i] == keys[i])) *err = 1; /* partial_cmp(..).unwrap() panics on NaN (bvh.rs:99,109,119) */
    }
    if (!*err) stable_sort_by_key(objects, keys, n, tmp_o, tmp_k);
    free(keys); free(tmp_k); free(tmp_o);

    int nl = n / 2;
    uint32_t *left = (uint32_t *)malloc((size_t)(nl > 0 ? nl : 1) * sizeof *left);
    uint32_t *right = (uint32_t *)malloc((size_t)(n - nl) * sizeof *right);
    memcpy(left, objects, (size_t)nl * sizeof *left);
    memcpy(right, objects + nl, (size_t)(n - nl) * sizeof *right);
    node->kind = 2;
    node->bounds = bounds;
    node->left = bvh_from_list(left, nl, coll, err);
    node->right = bvh_from_list(right, n - nl, coll, err);
    free(left); free(right);
    return node;
}

/* bvh.rs:146-172 BvhNode::hit + bvh.rs:177-199 merge_optionals */
static int bvh_hit(const bvh_node *n, const ray_t *ray, float tmin, float tmax, const collection *coll, hit_t *out,
                   tctx_t *tc) {
    if (n->kind == 1) return coll_hit(coll, n->handle, ray, tmin, tmax, out, tc);
    if (n->kind == 2) {
        tc->box_tests++;
        if (aabb_hit(&n->bounds, ray, tmin, tmax)) {
            hit_t hl, hr;
            int l = bvh_hit(n->left, ray, tmin, tmax, coll, &hl, tc);
            int r = bvh_hit(n->right, ray, tmin, tmax, coll, &hr, tc);
            if (l && r) { *out = (hl.t < hr.t) ? hl : hr; return 1; } /* tie or NaN -> right */
            if (l) { *out = hl; return 1; }
            if (r) { *out = hr; return 1; }
        }
    }
    return 0;
}

static inline float or_log(float x, int mode);
/* self.boundary.hit(ray, tmin, tmax) for the two kinds of boundary a Volume is built over here (volume.rs:35-36) */
static int volume_hit(const volume_t *v, const ray_t *ray, float tmin, float tmax, hit_t *out, tctx_t *tc, int trig_mode);
static int volume_boundary_hit(const volume_t *v, const ray_t *ray, float tmin, float tmax, hit_t *out, tctx_t *tc, int trig_mode) {
    if (v->boundary_volume) return volume_hit(v->boundary_volume, ray, tmin, tmax, out, tc, trig_mode); /* Hittable for Volume again */
    if (v->boundary_mesh) { /* Mesh::hit, mesh.rs:124-127: the mesh's own tree with the caller's range; Triangle::hit
                             * takes tmin as its determinant epsilon and its lower bound on t, and ignores tmax */
        collection c = {NULL, v->boundary_mesh};
        return bvh_hit(v->boundary_mesh->bvh_root, ray, tmin, tmax, &c, out, tc);
    }
    return sphere_hit(&v->boundary, ray, tmin, tmax, out, tc);
}
/* lib/volume.rs:34-73  Volume::hit */
static int volume_hit(const volume_t *v, const ray_t *ray, float tmin, float tmax, hit_t *out, tctx_t *tc, int trig_mode) {
    hit_t hit1, hit2;
    if (!volume_boundary_hit(v, ray, -INFINITY, INFINITY, &hit1, tc, trig_mode)) return 0;
    if (!volume_boundary_hit(v, ray, hit1.t + tmin, INFINITY, &hit2, tc, trig_mode)) return 0;
    if (hit1.t < tmin) hit1.t = tmin;
    if (hit2.t > tmax) hit2.t = tmax;
    if (hit1.t >= hit2.t) return 0;
    if (hit1.t < 0.0f) hit1.t = 0.0f;
    float ray_length = v3_mag(ray->direction);
    float distance_inside_boundary = (hit2.t - hit1.t) * ray_length;
    float random_float = or_rng_unit_inclusive(&tc->rng); /* gen_range(0.0..=1.0) */
    float hit_distance = v->neg_inv_density * or_log(random_float, trig_mode);
    if (hit_distance > distance_inside_boundary) return 0;
    float t = hit1.t + hit_distance / ray_length;
    *out = hit_new(ray_at(ray, t), v3_new(1.0f, 0.0f, 0.0f), t, ray, volume_material(v));
    return 1;
}

static int object_hit(const or_scene *sc, const object_t *o, const ray_t *ray, float tmin, float tmax, hit_t *out,
                      tctx_t *tc) {
    switch (o->kind) {
    case OBJ_SPHERE: return sphere_hit(&o->sphere, ray, tmin, tmax, out, tc);
    case OBJ_MESH: { /* mesh.rs:124-127 */
        collection c = {NULL, o->mesh};
        return bvh_hit(o->mesh->bvh_root, ray, tmin, tmax, &c, out, tc);
    }
    default: return volume_hit(&o->volume, ray, tmin, tmax, out, tc, sc->trig_mode);
    }
}

/* scene.rs:141-157  Hittable for Scene (linear world) */
static int scene_hit_linear(const or_scene *sc, const ray_t *ray, float tmin, float tmax, hit_t *out, tctx_t *tc) {
    int found = 0;
    float closest_so_far = tmax;
    for (int i = 0; i < sc->n_objects; i++) {
        hit_t h;
        if (object_hit(sc, &sc->objects[i], ray, tmin, closest_so_far, &h, tc)) {
            closest_so_far = h.t;
            *out = h;
            found = 1;
        }
    }
    return found;
}

typedef struct { const or_scene *scene; const bvh_node *bvh; } world_t; /* render.rs:67-72 */
static int world_hit(const world_t *w, const ray_t *ray, float tmin, float tmax, hit_t *out, tctx_t *tc) {
    if (w->bvh) { /* BvhTree::hit bvh.rs:43-50 */
        collection c = {w->scene, NULL};
        return bvh_hit(w->bvh, ray, tmin, tmax, &c, out, tc);
    }
    return scene_hit_linear(w->scene, ray, tmin, tmax, out, tc);
}

/* ------------------------------------------------------------------------------------------ */
/* lib/utils/math.rs */
static inline int near_zero(v3 v) { /* math.rs:6-9 */
    const float EPS = 1.0e-8f;
    return fabsf(v.x) < EPS && fabsf(v.y) < EPS && fabsf(v.z) < EPS;
}
static inline v3 reflect(v3 v, v3 n) { return v3_sub(v, v3_scale(n, 2.0f * v3_dot(v, n))); } /* math.rs:12-14 */
static inline v3 refract(v3 uv, v3 n, float etai_over_etat) {                                  /* math.rs:16-21 */
    float cos_theta = fminf(v3_dot(n, v3_neg(uv)), 1.0f);
    v3 out_perp = v3_scale(v3_add(uv, v3_scale(n, cos_theta)), etai_over_etat);
    v3 out_parallel = v3_scale(n, -sqrtf(fabsf(1.0f - v3_mag2(out_perp))));
    return v3_add(out_perp, out_parallel);
}

/* lib/utils/sample.rs:26-79 */
static v3 sample_unit_sphere_volume(or_rng *rng) { /* sample.rs:43-51 */
    uint32_t s = or_rng_u32(rng); /* one word of the path's stream per call, whatever the number of rejections */
    for (;;) {
        float x = or_ball_coord(&s), y = or_ball_coord(&s), z = or_ball_coord(&s);
        v3 v = v3_new(x, y, z);
        if (v3_mag2(v) <= 1.0f) return v;
    }
}
static v3 sample_unit_sphere_surface(or_rng *rng) { return v3_normalize(sample_unit_sphere_volume(rng)); } /* :66-70 */

/* the polynomial acos/atan2 the HIP kernels use (DESIGN.md "transcendentals"); cephes coefficients */
static float poly_asin_core(float x) {
    float z = x * x;
    float p = 4.2163199048E-2f;
    p = p * z + 2.4181311049E-2f;
    p = p * z + 4.5470025998E-2f;
    p = p * z + 7.4953002686E-2f;
    p = p * z + 1.6666752422E-1f;
    return (p * z) * x + x;
}
static float poly_acos(float x) {
    float a = fabsf(x);
    if (a <= 0.5f) return 1.57079632679489661923f - poly_asin_core(x);
    float z = (1.0f - a) * 0.5f;
    float r = 2.0f * poly_asin_core(sqrtf(z)); /* |x|>1 -> sqrt(negative) = NaN, as acosf */
    return x > 0.0f ? r : OR_PI - r;
}
static float poly_atan_pos(float x) {
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
static float poly_atan2(float y, float x) {
    if (!(x == x) || !(y == y)) return NAN;
    if (y == 0.0f) return (x < 0.0f || (x == 0.0f && signbit(x))) ? copysignf(OR_PI, y) : y;
    if (x == 0.0f) return y > 0.0f ? 1.57079632679489661923f : -1.57079632679489661923f;
    float r = poly_atan_pos(fabsf(y) / fabsf(x));
    if (x < 0.0f) r = OR_PI - r;
    return y < 0.0f ? -r : r;
}
/* cephes-style logf, the sequence the HIP kernels use for Volume::hit's ln (volume.rs:58) */
static float poly_log(float xx) {
    if (!(xx > 0.0f)) return xx == 0.0f ? -INFINITY : NAN;
    if (xx == INFINITY) return xx;
    uint32_t b;
    memcpy(&b, &xx, 4);
    int e = (int)((b >> 23) & 0xffu) - 126;
    if (((b >> 23) & 0xffu) == 0u) return logf(xx); /* denormal: never produced by the 23-bit uniform */
    b = (b & 0x007fffffu) | 0x3f000000u;
    float x;
    memcpy(&x, &b, 4); /* frexp mantissa in [0.5, 1) */
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
    float fe = (float)e;
    if (e) y += -2.12194440e-4f * fe;
    y += -0.5f * z;
    z = x + y;
    if (e) z += 0.693359375f * fe;
    return z;
}
static inline float or_log(float x, int mode) { return mode == OR_TRIG_POLY ? poly_log(x) : logf(x); }
static inline float or_acos(float x, int mode) { return mode == OR_TRIG_POLY ? poly_acos(x) : acosf(x); }
static inline float or_atan2(float y, float x, int mode) { return mode == OR_TRIG_POLY ? poly_atan2(y, x) : atan2f(y, x); }

/* ------------------------------------------------------------------------------------------ */
/* lib/materials.rs: Material::scatter -> (attenuation, Option<Ray>)                            */
static float reflectance(float cosine, float idx) { /* materials.rs:103-108 */
    float r0 = (1.0f - idx) / (1.0f + idx);
    r0 = r0 * r0;
    float x = 1.0f - cosine; /* powi(x,5): compiler-rt __powisf2 = x * ((x*x)*(x*x)) */
    float x2 = x * x;
    float x4 = x2 * x2;
    return r0 + (1.0f - r0) * (x * x4);
}
static int material_scatter(const or_scene *sc, int mi, const ray_t *ray, const hit_t *hit, col *att, ray_t *scattered,
                            tctx_t *tc) {
    const material_t *m = &sc->materials[mi];
    switch (m->kind) {
    case OR_MAT_LAMBERTIAN: { /* materials.rs:31-41 */
        v3 dir = v3_add(hit->normal, sample_unit_sphere_surface(&tc->rng));
        if (near_zero(dir)) dir = hit->normal;
        scattered->origin = hit->point;
        scattered->direction = dir;
        *att = m->color;
        return 1;
    }
    case OR_MAT_METAL: { /* materials.rs:58-70 */
        v3 reflected = v3_normalize(reflect(ray->direction, hit->normal));
        scattered->origin = hit->point;
        scattered->direction = v3_add(reflected, v3_scale(sample_unit_sphere_surface(&tc->rng), m->param));
        if (v3_dot(scattered->direction, hit->normal) > 0.0f) { *att = m->color; return 1; }
        *att = col_new(0.0f, 0.0f, 0.0f);
        return 0;
    }
    case OR_MAT_EMISSION: /* materials.rs:86-91 */
        *att = m->color;
        return 0;
    case OR_MAT_DIELECTRIC: { /* materials.rs:111-138 */
        float refraction_ratio = hit->front_face ? 1.0f / m->param : m->param;
        v3 unit_direction = v3_normalize(ray->direction);
        float cos_theta = fminf(v3_dot(hit->normal, v3_neg(unit_direction)), 1.0f);
        float sin_theta = sqrtf(1.0f - cos_theta * cos_theta);
        int cannot_refract = (refraction_ratio * sin_theta) > 1.0f;
        v3 direction;
        /* short-circuit ||: no draw on total internal reflection */
        if (cannot_refract || reflectance(cos_theta, refraction_ratio) > or_rng_unit(&tc->rng))
            direction = reflect(unit_direction, hit->normal);
        else
            direction = refract(unit_direction, hit->normal, refraction_ratio);
        *att = col_new(1.0f, 1.0f, 1.0f);
        scattered->origin = hit->point;
        scattered->direction = direction;
        return 1;
    }
    case OR_MAT_MIX: /* materials.rs:148-157 */
        if (or_rng_unit(&tc->rng) >= m->param) return material_scatter(sc, m->first, ray, hit, att, scattered, tc);
        return material_scatter(sc, m->second, ray, hit, att, scattered, tc);
    default: /* Isotropic, volume.rs:80-88 */
        *att = m->color;
        scattered->origin = hit->point;
        scattered->direction = sample_unit_sphere_volume(&tc->rng);
        return 1;
    }
}

/* lib/backgrounds.rs: Background::sample */
static col background_sample(const or_scene *sc, const ray_t *ray, tctx_t *tc) {
    switch (sc->bg_kind) {
    case 0: return sc->bg_a; /* backgrounds.rs:25-30 */
    case 1: {                /* backgrounds.rs:44-49 + ray.rs:19-26 */
        float t = 0.5f * (v3_normalize(ray->direction).y + 1.0f);
        return col_add(col_scale(sc->bg_a, 1.0f - t), col_scale(sc->bg_b, t));
    }
    default: { /* backgrounds.rs:84-92 + math.rs:24-29 */
        tc->sky_lookups++;
        v3 d = v3_normalize(ray->direction);
        float sx = or_acos(-d.y, sc->trig_mode);
        float sy = or_atan2(-d.z, d.x, sc->trig_mode) + OR_PI;
        float u = sx / OR_PI;
        float v = sy / (2.0f * OR_PI);
        if (sc->sky_bilinear) { /* opt-in extension, the same arithmetic as csrc/rtr_kernels.hip sky_bilinear */
            float fx = v * (float)sc->sky_w - 0.5f, fy = u * (float)sc->sky_h - 0.5f;
            if (!(fx == fx)) fx = 0.0f;
            if (!(fy == fy)) fy = 0.0f;
            float x0f = floorf(fx), r0f = floorf(fy);
            float tx = fx - x0f, ty = fy - r0f;
            int64_t w = (int64_t)sc->sky_w, h = (int64_t)sc->sky_h;
            int64_t x0 = (int64_t)x0f, r0 = (int64_t)r0f;
            int64_t x1 = x0 + 1, r1 = r0 + 1;
            x0 = ((x0 % w) + w) % w; x1 = ((x1 % w) + w) % w;           /* longitude wraps */
            r0 = r0 < 0 ? 0 : (r0 > h - 1 ? h - 1 : r0); r1 = r1 < 0 ? 0 : (r1 > h - 1 ? h - 1 : r1); /* latitude clamps */
            int64_t y0 = h - 1 - r0, y1 = h - 1 - r1;
            col t00 = sc->sky[x0 + y0 * w], t01 = sc->sky[x1 + y0 * w], t10 = sc->sky[x0 + y1 * w], t11 = sc->sky[x1 + y1 * w];
            float ux = 1.0f - tx, uy = 1.0f - ty;
            col a = col_add(col_scale(t00, ux), col_scale(t01, tx));
            col b = col_add(col_scale(t10, ux), col_scale(t11, tx));
            return col_add(col_scale(a, uy), col_scale(b, ty));
        }
        uint64_t x = sat_usize(v * (float)sc->sky_w) % sc->sky_w;
        uint64_t y = sc->sky_h - 1 - sat_usize(u * (float)sc->sky_h) % sc->sky_h;
        return sc->sky[x + y * sc->sky_w];
    }
    }
}

/* lib/core/render.rs:32-59  trace_ray */
static col trace_ray(const or_scene *sc, const world_t *world, const ray_t *ray, uint32_t depth, tctx_t *tc) {
    if (depth >= sc->max_ray_depth) return col_new(0.0f, 0.0f, 0.0f);
    hit_t hit;
    tc->segments++;
    if (world_hit(world, ray, 0.00001f, INFINITY, &hit, tc)) {
        col attenuation;
        ray_t scattered;
        int has;
        switch (sc->mode) {
        case OR_MODE_FULL: has = material_scatter(sc, hit.material, ray, &hit, &attenuation, &scattered, tc); break;
        case OR_MODE_CLAY: { /* Lambertian::new(0.8,0.8,0.8).scatter */
            v3 dir = v3_add(hit.normal, sample_unit_sphere_surface(&tc->rng));
            if (near_zero(dir)) dir = hit.normal;
            scattered.origin = hit.point;
            scattered.direction = dir;
            attenuation = col_new(0.8f, 0.8f, 0.8f);
            has = 1;
            break;
        }
        case OR_MODE_NORMAL: {
            v3 n = v3_scale(v3_add(v3_normalize(hit.normal), v3_new(1.0f, 1.0f, 1.0f)), 0.5f);
            return col_new(n.x, n.y, n.z);
        }
        default: return col_new(0.0f, 0.0f, 0.0f);
        }
        if (has) return col_mul(attenuation, trace_ray(sc, world, &scattered, depth + 1, tc));
        return attenuation;
    }
    return background_sample(sc, ray, tc);
}

/* lib/core/camera.rs:37-56 */
typedef struct { v3 origin, horizontal, vertical, lower_left_corner; } ray_origin_t;
static ray_origin_t camera_ray_origin(const or_scene *s) {
    float theta = degrees_to_radians(s->vertical_fov);
    float h = tanf(theta / 2.0f);
    float viewport_height = 2.0f * h;
    float viewport_width = s->aspect_ratio * viewport_height;
    v3 w = v3_normalize(v3_sub(s->lookfrom, s->lookat));
    v3 u = v3_normalize(v3_cross(s->vertical, w));
    v3 v = v3_cross(w, u);
    ray_origin_t ro;
    ro.origin = s->lookfrom;
    ro.horizontal = v3_scale(u, viewport_width);
    ro.vertical = v3_scale(v, viewport_height);
    ro.lower_left_corner =
        v3_sub(v3_add(v3_sub(s->lookfrom, v3_divs(ro.horizontal, 2.0f)), v3_divs(ro.vertical, 2.0f)), w);
    return ro;
}
/* camera.rs:19-24 */
static inline ray_t get_ray(const ray_origin_t *ro, float s, float t) {
    ray_t r;
    r.origin = ro->origin;
    r.direction = v3_sub(v3_sub(v3_add(ro->lower_left_corner, v3_scale(ro->horizontal, s)), v3_scale(ro->vertical, t)),
                         ro->origin);
    return r;
}

/* ------------------------------------------------------------------------------------------ */
/* render(), render.rs:61-127                                                                   */
typedef struct {
    const or_scene *scene;
    world_t world;
    ray_origin_t ro;
    uint32_t width, height, row_begin, row_end;
    uint64_t seed;
    uint8_t *rgba8;
    float *mean_rgb;
    atomic_uint next_row;
    tctx_t totals;
    pthread_mutex_t lock;
} job_t;

/* the closure calculate_pixel, render.rs:75-111 */
static void calculate_pixel(job_t *job, uint32_t index, tctx_t *tc) {
    const or_scene *sc = job->scene;
    uint32_t x = index % job->width;
    uint32_t y = index / job->width;
    col color = col_new(0.0f, 0.0f, 0.0f);
    for (uint32_t s = 0; s < sc->samples_per_pixel; s++) {
        or_rng_seed(&tc->rng, job->seed, index, s);
        float u = ((float)x + or_rng_unit_inclusive(&tc->rng)) / (float)(job->width - 1);
        float v = ((float)y + or_rng_unit_inclusive(&tc->rng)) / (float)(job->height - 1);
        ray_t ray = get_ray(&job->ro, u, v);
        col rc = trace_ray(sc, &job->world, &ray, 0, tc);
        /* f32::clamp(0, clamp_indirect): NaN stays NaN (render.rs:95-97) */
        float lo = 0.0f, hi = sc->clamp_indirect;
        if (rc.r < lo) rc.r = lo; if (rc.r > hi) rc.r = hi;
        if (rc.g < lo) rc.g = lo; if (rc.g > hi) rc.g = hi;
        if (rc.b < lo) rc.b = lo; if (rc.b > hi) rc.b = hi;
        color = col_add(color, rc);
    }
    float n = (float)sc->samples_per_pixel;
    col mean = col_new(color.r / n, color.g / n, color.b / n);
    if (job->mean_rgb) {
        job->mean_rgb[3 * (size_t)index + 0] = mean.r;
        job->mean_rgb[3 * (size_t)index + 1] = mean.g;
        job->mean_rgb[3 * (size_t)index + 2] = mean.b;
    }
    if (job->rgba8) {
        col g = col_new(sqrtf(mean.r), sqrtf(mean.g), sqrtf(mean.b));
        uint8_t *p = job->rgba8 + 4 * (size_t)index; /* color.rs:34-41 into_raw */
        p[0] = sat_u8(g.r * 255.0f);
        p[1] = sat_u8(g.g * 255.0f);
        p[2] = sat_u8(g.b * 255.0f);
        p[3] = 255;
    }
}

static void *worker(void *arg) {
    job_t *job = (job_t *)arg;
    tctx_t tc;
    memset(&tc, 0, sizeof tc);
    for (;;) {
        uint32_t row = job->row_begin + atomic_fetch_add(&job->next_row, 1u);
        if (row >= job->row_end) break;
        for (uint32_t x = 0; x < job->width; x++) calculate_pixel(job, row * job->width + x, &tc);
    }
    tc.draws = tc.rng.draws;
    pthread_mutex_lock(&job->lock);
    job->totals.segments += tc.segments;
    job->totals.box_tests += tc.box_tests;
    job->totals.sphere_tests += tc.sphere_tests;
    job->totals.tri_tests += tc.tri_tests;
    job->totals.sky_lookups += tc.sky_lookups;
    job->totals.draws += tc.draws;
    pthread_mutex_unlock(&job->lock);
    return NULL;
}

static bvh_node *scene_build_bvh(const or_scene *s, int *err) { /* scene.rs:112-115, bvh.rs:35-40 */
    uint32_t *objs = (uint32_t *)malloc((size_t)(s->n_objects > 0 ? s->n_objects : 1) * sizeof *objs);
    for (int i = 0; i < s->n_objects; i++) objs[i] = (uint32_t)i;
    collection c = {s, NULL};
    bvh_node *root = bvh_from_list(objs, s->n_objects, &c, err);
    free(objs);
    return root;
}

static int validate(const or_scene *s) {
    for (int i = 0; i < s->n_objects; i++) {
        const object_t *o = &s->objects[i];
        int m = o->kind == OBJ_SPHERE ? o->sphere.material : (o->kind == OBJ_MESH ? o->mesh->material : volume_material(&o->volume));
        if (m < 0 || m >= s->n_materials) return or_fail(-3, "material index out of bounds (scene.rs:106-110 would panic)");
    }
    if (!(s->clamp_indirect == s->clamp_indirect) || s->clamp_indirect < 0.0f)
        return or_fail(-4, "clamp_indirect NaN or negative (f32::clamp panics, render.rs:95)");
    return 0;
}

int or_render(or_scene *s, uint32_t width, uint32_t height, uint64_t seed, int threads, uint32_t row_begin,
              uint32_t row_end, uint8_t *rgba8, float *mean_rgb, or_stats *stats) {
    int rc = validate(s);
    if (rc) return rc;
    if (width == 0 || height == 0) return or_fail(-5, "empty render target");
    struct timespec t0, t1;
    clock_gettime(CLOCK_MONOTONIC, &t0);

    job_t job;
    memset(&job, 0, sizeof job);
    job.scene = s;
    job.ro = camera_ray_origin(s);
    int err = 0;
    bvh_node *bvh = scene_build_bvh(s, &err); /* always built, render.rs:67 */
    if (err) { bvh_free(bvh); return or_fail(-2, "NaN centroid in BVH build (partial_cmp().unwrap() panics)"); }
    job.world.scene = s;
    job.world.bvh = s->enable_bvh_tree ? bvh : NULL;
    job.width = width; job.height = height;
    if (row_end > row_begin) { job.row_begin = row_begin; job.row_end = row_end < height ? row_end : height; }
    else { job.row_begin = 0; job.row_end = height; }
    job.seed = seed; job.rgba8 = rgba8; job.mean_rgb = mean_rgb;
    atomic_init(&job.next_row, 0u);
    pthread_mutex_init(&job.lock, NULL);

    int nthreads = threads;
    if (nthreads <= 0) nthreads = s->enable_multithreading ? (int)sysconf(_SC_NPROCESSORS_ONLN) : 1;
    if (nthreads < 1) nthreads = 1;
    if (nthreads > 256) nthreads = 256;
    if (nthreads == 1) {
        worker(&job);
    } else {
        pthread_t th[256];
        for (int i = 0; i < nthreads; i++) pthread_create(&th[i], NULL, worker, &job);
        for (int i = 0; i < nthreads; i++) pthread_join(th[i], NULL);
    }
    bvh_free(bvh);
    pthread_mutex_destroy(&job.lock);
    clock_gettime(CLOCK_MONOTONIC, &t1);
    if (stats) {
        memset(stats, 0, sizeof *stats);
        stats->primary_samples = (uint64_t)width * (job.row_end - job.row_begin) * s->samples_per_pixel;
        stats->segments = job.totals.segments;
        stats->box_tests = job.totals.box_tests;
        stats->sphere_tests = job.totals.sphere_tests;
        stats->tri_tests = job.totals.tri_tests;
        stats->sky_lookups = job.totals.sky_lookups;
        stats->rng_draws = job.totals.draws;
        stats->seconds = (double)(t1.tv_sec - t0.tv_sec) + 1e-9 * (double)(t1.tv_nsec - t0.tv_nsec);
        stats->threads = nthreads;
    }
    return 0;
}

/* ------------------------------------------------------------------------------------------ */
/* scene construction                                                                           */
or_scene *or_scene_new(void) { /* Scene::default(), scene.rs:42-65 */
    or_scene *s = (or_scene *)calloc(1, sizeof *s);
    s->lookfrom = v3_new(0, 0, 0); s->lookat = v3_new(0, 0, -1); s->vertical = v3_new(0, 1, 0);
    s->vertical_fov = 90.0f; s->aspect_ratio = 1.0f;
    s->samples_per_pixel = 5; s->max_ray_depth = 6; s->clamp_indirect = 10.0f;
    s->enable_multithreading = 1; s->enable_bvh_tree = 1; s->mode = OR_MODE_FULL;
    s->bg_kind = 0; s->bg_a = col_new(0.8f, 0.8f, 0.8f);
    return s;
}
static void mesh_free(mesh_t *m) {
    if (!m) return;
    free(m->vertices); free(m->triangles); bvh_free(m->bvh_root); free(m);
}
void or_scene_free(or_scene *s) {
    if (!s) return;
    for (int i = 0; i < s->n_objects; i++) {
        if (s->objects[i].kind == OBJ_MESH) mesh_free(s->objects[i].mesh);
        if (s->objects[i].kind == OBJ_VOLUME) {
            volume_t *v = &s->objects[i].volume, *inner = v->boundary_volume;
            if (v->boundary_mesh) mesh_free(v->boundary_mesh);
            while (inner) { /* the chain of boxed boundary Volumes */
                volume_t *next = inner->boundary_volume;
                if (inner->boundary_mesh) mesh_free(inner->boundary_mesh);
                free(inner);
                inner = next;
            }
        }
    }
    free(s->objects); free(s->materials); free(s->sky); free(s);
}
void or_set_camera(or_scene *s, const float lookfrom[3], const float lookat[3], const float vertical[3],
                   float vertical_fov, float aspect_ratio) {
    s->lookfrom = v3_new(lookfrom[0], lookfrom[1], lookfrom[2]);
    s->lookat = v3_new(lookat[0], lookat[1], lookat[2]);
    s->vertical = v3_new(vertical[0], vertical[1], vertical[2]);
    s->vertical_fov = vertical_fov; s->aspect_ratio = aspect_ratio;
}
void or_set_settings(or_scene *s, uint32_t spp, uint32_t max_ray_depth, float clamp_indirect, int mt, int bvh, int mode) {
    s->samples_per_pixel = spp; s->max_ray_depth = max_ray_depth > 255 ? 255 : max_ray_depth; /* u8, scene.rs:25 */
    s->clamp_indirect = clamp_indirect; s->enable_multithreading = mt; s->enable_bvh_tree = bvh; s->mode = mode;
}
void or_set_trig_mode(or_scene *s, int m) { s->trig_mode = m; }
void or_set_sky_filter(or_scene *s, int bilinear) { s->sky_bilinear = bilinear ? 1 : 0; }
void or_set_background_uniform(or_scene *s, const float rgb[3]) { s->bg_kind = 0; s->bg_a = col_new(rgb[0], rgb[1], rgb[2]); }
void or_set_background_gradient(or_scene *s, const float top[3], const float bottom[3]) {
    s->bg_kind = 1; s->bg_a = col_new(top[0], top[1], top[2]); s->bg_b = col_new(bottom[0], bottom[1], bottom[2]);
}
int or_set_background_skymap(or_scene *s, const float *image_rgb, uint32_t width, uint32_t height) {
    if (!image_rgb || width == 0 || height == 0) return or_fail(-6, "empty sky image (% width would panic)");
    free(s->sky);
    size_t n = (size_t)width * height;
    s->sky = (col *)malloc(n * sizeof *s->sky);
    for (size_t i = 0; i < n; i++) s->sky[i] = col_new(image_rgb[3 * i], image_rgb[3 * i + 1], image_rgb[3 * i + 2]);
    s->sky_w = width; s->sky_h = height; s->bg_kind = 2;
    return 0;
}
int or_add_material(or_scene *s, int kind, const float rgb[3], float param, int first, int second) {
    if (kind == OR_MAT_MIX && (first < 0 || second < 0 || first >= s->n_materials || second >= s->n_materials))
        return or_fail(-7, "mix children must be previously added materials");
    if (s->n_materials == s->cap_materials) {
        s->cap_materials = s->cap_materials ? 2 * s->cap_materials : 16;
        s->materials = (material_t *)realloc(s->materials, (size_t)s->cap_materials * sizeof *s->materials);
    }
    material_t *m = &s->materials[s->n_materials];
    m->kind = kind; m->color = rgb ? col_new(rgb[0], rgb[1], rgb[2]) : col_new(0, 0, 0);
    m->param = param; m->first = first; m->second = second;
    return s->n_materials++;
}
static object_t *push_object(or_scene *s) {
    if (s->n_objects == s->cap_objects) {
        s->cap_objects = s->cap_objects ? 2 * s->cap_objects : 16;
        s->objects = (object_t *)realloc(s->objects, (size_t)s->cap_objects * sizeof *s->objects);
    }
    object_t *o = &s->objects[s->n_objects++];
    memset(o, 0, sizeof *o);
    return o;
}
int or_add_sphere(or_scene *s, const float c[3], float radius, int material) {
    object_t *o = push_object(s);
    o->kind = OBJ_SPHERE; o->sphere.center = v3_new(c[0], c[1], c[2]); o->sphere.radius = radius; o->sphere.material = material;
    return s->n_objects - 1;
}
int or_add_volume_sphere(or_scene *s, const float c[3], float radius, int material, float density) {
    object_t *o = push_object(s);
    o->kind = OBJ_VOLUME;
    o->volume.boundary.center = v3_new(c[0], c[1], c[2]); o->volume.boundary.radius = radius;
    o->volume.boundary.material = material;
    o->volume.neg_inv_density = -1.0f / density; /* volume.rs:26-31 */
    return s->n_objects - 1;
}
/* Mesh::from_buffers + build_bvh (mesh.rs:78-121); NULL (with the error set) on failure */
static mesh_t *mesh_build(const float *positions, const float *normals, uint32_t n_verts, const uint32_t *indices,
                          uint32_t n_tris, int material) {
    for (uint32_t i = 0; i < 3 * n_tris; i++)
        if (indices[i] >= n_verts) { or_fail(-8, "vertex index out of bounds (mesh.rs:84-88 would panic)"); return NULL; }
    mesh_t *m = (mesh_t *)calloc(1, sizeof *m);
    m->n_verts = n_verts; m->n_tris = n_tris; m->material = material;
    m->vertices = (vertex_t *)malloc((size_t)(n_verts ? n_verts : 1) * sizeof *m->vertices);
    m->triangles = (triangle_t *)malloc((size_t)(n_tris ? n_tris : 1) * sizeof *m->triangles);
    for (uint32_t i = 0; i < n_verts; i++) {
        m->vertices[i].position = v3_new(positions[3 * i], positions[3 * i + 1], positions[3 * i + 2]);
        m->vertices[i].normal = v3_new(normals[3 * i], normals[3 * i + 1], normals[3 * i + 2]);
    }
    for (uint32_t t = 0; t < n_tris; t++) { /* mesh.rs:85-96 */
        const uint32_t *tri = indices + 3 * t;
        v3 e1 = v3_sub(m->vertices[tri[0]].position, m->vertices[tri[1]].position);
        v3 e2 = v3_sub(m->vertices[tri[2]].position, m->vertices[tri[1]].position);
        m->triangles[t].normal = v3_normalize(v3_cross(e2, e1));
        m->triangles[t].v[0] = tri[0]; m->triangles[t].v[1] = tri[1]; m->triangles[t].v[2] = tri[2];
    }
    aabb_t bounds = aabb_default(); /* mesh.rs:99-108: seeded with the zero box */
    for (uint32_t i = 0; i < n_verts; i++) bounds = aabb_surround(bounds, aabb_from_point(m->vertices[i].position));
    m->bounds = bounds;
    /* Mesh::build_bvh, mesh.rs:119-121 */
    uint32_t *objs = (uint32_t *)malloc((size_t)(n_tris ? n_tris : 1) * sizeof *objs);
    for (uint32_t i = 0; i < n_tris; i++) objs[i] = i;
    collection c = {NULL, m};
    int err = 0;
    m->bvh_root = bvh_from_list(objs, (int)n_tris, &c, &err);
    free(objs);
    if (err) { mesh_free(m); or_fail(-2, "NaN centroid in mesh BVH build"); return NULL; }
    return m;
}
int or_add_mesh(or_scene *s, const float *positions, const float *normals, uint32_t n_verts, const uint32_t *indices,
                uint32_t n_tris, int material) {
    mesh_t *m = mesh_build(positions, normals, n_verts, indices, n_tris, material);
    if (!m) return -1;
    object_t *o = push_object(s);
    o->kind = OBJ_MESH; o->mesh = m;
    return s->n_objects - 1;
}
/* Volume::new(Box::new(mesh), density), volume.rs:25-32 */
int or_add_volume_mesh(or_scene *s, const float *positions, const float *normals, uint32_t n_verts, const uint32_t *indices,
                       uint32_t n_tris, int material, float density) {
    mesh_t *m = mesh_build(positions, normals, n_verts, indices, n_tris, material);
    if (!m) return -1;
    object_t *o = push_object(s);
    o->kind = OBJ_VOLUME;
    o->volume.boundary_mesh = m;
    o->volume.neg_inv_density = -1.0f / density;
    return s->n_objects - 1;
}
/* Volume::new(Box::new(<the Volume added last>), density): the scene's last object, which must be a Volume, becomes the
 * boundary of a new Volume that takes its place (volume.rs:20 admits any Object as a boundary) */
int or_wrap_last_volume(or_scene *s, float density) {
    if (s->n_objects < 1 || s->objects[s->n_objects - 1].kind != OBJ_VOLUME) return or_fail(-9, "the last object is not a Volume");
    object_t *o = &s->objects[s->n_objects - 1];
    volume_t *inner = (volume_t *)malloc(sizeof *inner);
    *inner = o->volume;
    memset(&o->volume, 0, sizeof o->volume);
    o->volume.boundary_volume = inner;
    o->volume.neg_inv_density = -1.0f / density;
    return s->n_objects - 1;
}

/* ------------------------------------------------------------------------------------------ */
/* probes for the known-answer tests                                                            */
void or_camera_ray_origin(const or_scene *s, float out[12]) {
    ray_origin_t ro = camera_ray_origin(s);
    const v3 *v[4] = {&ro.origin, &ro.horizontal, &ro.vertical, &ro.lower_left_corner};
    for (int i = 0; i < 4; i++) { out[3 * i] = v[i]->x; out[3 * i + 1] = v[i]->y; out[3 * i + 2] = v[i]->z; }
}
int or_probe_hit(or_scene *s, const float origin[3], const float dir[3], float out[9]) {
    int err = 0;
    bvh_node *bvh = scene_build_bvh(s, &err);
    if (err) { bvh_free(bvh); return or_fail(-2, "NaN centroid"); }
    world_t w = {s, s->enable_bvh_tree ? bvh : NULL};
    ray_t ray = {v3_new(origin[0], origin[1], origin[2]), v3_new(dir[0], dir[1], dir[2])};
    tctx_t tc; memset(&tc, 0, sizeof tc);
    hit_t h;
    int hit = world_hit(&w, &ray, 0.00001f, INFINITY, &h, &tc);
    bvh_free(bvh);
    if (hit) {
        out[0] = h.t; out[1] = h.point.x; out[2] = h.point.y; out[3] = h.point.z;
        out[4] = h.normal.x; out[5] = h.normal.y; out[6] = h.normal.z; out[7] = (float)h.front_face; out[8] = (float)h.material;
    }
    return hit;
}
int or_probe_aabb(const float bmin[3], const float bmax[3], const float origin[3], const float dir[3], float tmin, float tmax) {
    aabb_t b = {v3_new(bmin[0], bmin[1], bmin[2]), v3_new(bmax[0], bmax[1], bmax[2])};
    ray_t ray = {v3_new(origin[0], origin[1], origin[2]), v3_new(dir[0], dir[1], dir[2])};
    return aabb_hit(&b, &ray, tmin, tmax);
}
static void leaf_order(const bvh_node *n, uint32_t *out, int cap, int *count) {
    if (!n) return;
    if (n->kind == 1) { if (out && *count < cap) out[*count] = n->handle; (*count)++; }
    else if (n->kind == 2) { leaf_order(n->left, out, cap, count); leaf_order(n->right, out, cap, count); }
}
int or_probe_bvh_leaf_order(or_scene *s, uint32_t *out, int cap) {
    int err = 0, count = 0;
    bvh_node *bvh = scene_build_bvh(s, &err);
    if (!err) leaf_order(bvh, out, cap, &count);
    bvh_free(bvh);
    return err ? or_fail(-2, "NaN centroid") : count;
}
int or_probe_mesh_leaf_order(or_scene *s, int object, uint32_t *out, int cap) {
    if (object < 0 || object >= s->n_objects || s->objects[object].kind != OBJ_MESH) return or_fail(-9, "not a mesh");
    int count = 0;
    leaf_order(s->objects[object].mesh->bvh_root, out, cap, &count);
    return count;
}
void or_probe_rng(uint64_t seed, uint32_t pixel, uint32_t sample, int kind, float *out, int n) {
    or_rng r; r.draws = 0;
    or_rng_seed(&r, seed, pixel, sample);
    for (int i = 0; i < n; i++)
        out[i] = kind == 0 ? or_rng_unit_inclusive(&r) : (kind == 1 ? or_rng_unit(&r) : 0.0f);
    if (kind == 2) { /* the coordinates the rejection sampler would test, in order */
        uint32_t s = or_rng_u32(&r);
        for (int i = 0; i < n; i++) out[i] = or_ball_coord(&s);
    }
    if (kind == 3) { /* n / 3 successive calls of sample_unit_sphere_volume on the stream: the accepted points */
        for (int i = 0; i + 2 < n; i += 3) {
            v3 v = sample_unit_sphere_volume(&r);
            out[i] = v.x; out[i + 1] = v.y; out[i + 2] = v.z;
        }
    }
}
float or_probe_acos(float x, int trig_mode) { return or_acos(x, trig_mode); }
float or_probe_atan2(float y, float x, int trig_mode) { return or_atan2(y, x, trig_mode); }
float or_probe_log(float x, int trig_mode) { return or_log(x, trig_mode); }
