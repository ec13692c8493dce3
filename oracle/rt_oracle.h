/*
 * rt_oracle.h -- CPU restatement of LevKruglyak/RayTracingRust's per-pixel radiance
 * integrator.  TEST INFRASTRUCTURE ONLY.
 *
 * This library is the parity checker and the "port" CPU baseline.  Only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it; nothing under
 * raytracingrust_amd/ (the product) links, imports or calls it.
 *
 * PARITY STATUS: "parity unpinned".  The reference ships no tests, golden vectors or
 * recorded outputs, cannot be built here (Rust; no cargo/rustc in the
 * image) and draws every random number from an OS-seeded thread_rng().  The restatement is
 * therefore pinned only by RNG-free known-answer tests derived from the reference source
 * (tests/test_oracle_kat.py); its random stream is our own counter-based generator that
 * reproduces rand 0.8's float *distributions*, not its values.
 *
 * All path:line citations are relative to /root/reference.
 */
#ifndef RT_ORACLE_H
#define RT_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* lib/materials.rs + lib/volume.rs material kinds */
enum {
    OR_MAT_LAMBERTIAN = 0, /* materials.rs:18-42  */
    OR_MAT_METAL = 1,      /* materials.rs:44-71  */
    OR_MAT_EMISSION = 2,   /* materials.rs:73-91  */
    OR_MAT_DIELECTRIC = 3, /* materials.rs:93-139 */
    OR_MAT_MIX = 4,        /* materials.rs:141-157 */
    OR_MAT_ISOTROPIC = 5   /* volume.rs:75-88 */
};

/* lib/core/scene.rs:14-20 */
enum { OR_MODE_FULL = 0, OR_MODE_CLAY = 1, OR_MODE_RANDOM = 2, OR_MODE_NORMAL = 3 };

/* transcendentals (SkyMap::sample's acos/atan2, Volume::hit's ln): 0 = host libm (what Rust's f32 methods call),
 * 1 = the polynomial forms the HIP kernels use (DESIGN.md "transcendentals") */
enum { OR_TRIG_LIBM = 0, OR_TRIG_POLY = 1 };

typedef struct or_scene or_scene;

typedef struct or_stats {
    uint64_t primary_samples; /* W*H*spp */
    uint64_t segments;        /* world.hit calls issued by trace_ray (render.rs:38) */
    uint64_t box_tests;       /* AABB::hit calls (aabb.rs:81) under the reference's unculled traversal */
    uint64_t sphere_tests;    /* Sphere::hit calls (objects.rs:22) */
    uint64_t tri_tests;       /* Triangle::hit calls (mesh.rs:146) */
    uint64_t sky_lookups;     /* SkyMap::sample calls (backgrounds.rs:84) */
    uint64_t rng_draws;
    double seconds;           /* wall time of the render call, scene BVH build included (render.rs:67) */
    int threads;
} or_stats;

or_scene *or_scene_new(void); /* Scene::default(), scene.rs:42-65 */
void or_scene_free(or_scene *s);

void or_set_camera(or_scene *s, const float lookfrom[3], const float lookat[3], const float vertical[3],
                   float vertical_fov, float aspect_ratio);
/* RenderSettings, scene.rs:22-30 */
void or_set_settings(or_scene *s, uint32_t samples_per_pixel, uint32_t max_ray_depth, float clamp_indirect,
                     int enable_multithreading, int enable_bvh_tree, int mode);
void or_set_trig_mode(or_scene *s, int trig_mode);
/* 0 (default): SkyMap::sample's nearest texel (backgrounds.rs:84-92).  1: bilinear interpolation, an opt-in extension of the
 * GPU library (option "sky_filter"), restated here only so that it can be checked */
void or_set_sky_filter(or_scene *s, int bilinear);

void or_set_background_uniform(or_scene *s, const float rgb[3]);                      /* backgrounds.rs:14-30 */
void or_set_background_gradient(or_scene *s, const float top[3], const float bottom[3]); /* backgrounds.rs:32-49 */
/* image: width*height RGB float triples, row 0 = top scanline (backgrounds.rs:51-81); copied */
int or_set_background_skymap(or_scene *s, const float *image_rgb, uint32_t width, uint32_t height);

/* Returns the new material's index (Scene::add_material, scene.rs:100-103).  For OR_MAT_MIX,
 * first/second are indices of previously added materials (the reference nests boxes; an index
 * pair is the same tree).  param = fuzz | ir | factor. */
int or_add_material(or_scene *s, int kind, const float rgb[3], float param, int first, int second);

/* Scene::add_object, scene.rs:87-90.  Return the object index or a negative error. */
int or_add_sphere(or_scene *s, const float center[3], float radius, int material);
/* Mesh::from_buffers + Mesh::build_bvh (mesh.rs:78-121).  positions/normals: n_verts*3 floats;
 * indices: n_tris*3 */
int or_add_mesh(or_scene *s, const float *positions, const float *normals, uint32_t n_verts,
                const uint32_t *indices, uint32_t n_tris, int material);
/* Volume::new over a sphere boundary (volume.rs:25-32) */
int or_add_volume_sphere(or_scene *s, const float center[3], float radius, int material, float density);
/* Volume over a mesh boundary (volume.rs:20 `boundary: Box<dyn Object>`): the mesh is built as or_add_mesh builds it */
int or_add_volume_mesh(or_scene *s, const float *positions, const float *normals, uint32_t n_verts, const uint32_t *indices,
                       uint32_t n_tris, int material, float density);
/* wraps the object added last (a Volume) into a new Volume of the given density: Volume::new(Box::new(volume), density) */
int or_wrap_last_volume(or_scene *s, float density);

/* render(), render.rs:61-127.  rgba8: W*H*4 bytes (may be NULL).  mean_rgb: W*H*3 floats holding
 * color/spp BEFORE the sqrt gamma (render.rs:103-107), may be NULL.  threads<=0: all cores when
 * enable_multithreading, else 1.  Rows [row_begin,row_end) only are rendered when row_end>row_begin
 * (bounded CPU-baseline samples); pass 0,0 for the full frame.  Returns 0 or a negative error
 * (reference panics become error codes). */
int or_render(or_scene *s, uint32_t width, uint32_t height, uint64_t seed, int threads, uint32_t row_begin,
              uint32_t row_end, uint8_t *rgba8, float *mean_rgb, or_stats *stats);

/* --- probes used by the known-answer tests ------------------------------------------------- */
/* Camera::ray_origin (camera.rs:37-56): out[12] = origin, horizontal, vertical, lower_left_corner */
void or_camera_ray_origin(const or_scene *s, float out[12]);
/* one world.hit (BVH or linear per settings): returns 1 on hit; out[9] = t, point, normal, front_face, material */
int or_probe_hit(or_scene *s, const float origin[3], const float dir[3], float out[9]);
/* AABB::hit (aabb.rs:81-137) */
int or_probe_aabb(const float bmin[3], const float bmax[3], const float origin[3], const float dir[3],
                  float tmin, float tmax);
/* scene-BVH leaves in depth-first order; returns count (<= cap); out may be NULL to query the count */
int or_probe_bvh_leaf_order(or_scene *s, uint32_t *out, int cap);
/* mesh-BVH leaves (triangle handles) of object `object` in depth-first order */
int or_probe_mesh_leaf_order(or_scene *s, int object, uint32_t *out, int cap);
/* the generator: n floats from stream (seed,pixel,sample); kind 0 = Uniform(0..=1), 1 = gen_range(0..1),
 * 2 = gen_range(-1..1) */
void or_probe_rng(uint64_t seed, uint32_t pixel, uint32_t sample, int kind, float *out, int n);
float or_probe_acos(float x, int trig_mode);
float or_probe_atan2(float y, float x, int trig_mode);
float or_probe_log(float x, int trig_mode);
const char *or_last_error(void);

#ifdef __cplusplus
}
#endif
#endif
