/*
 * rtr_hip.h -- C ABI of the MI355X-native path-tracing hot loop.
 *
 * This is the drop-in boundary for RayTracingRust's per-pixel radiance integrator.  The reference
 * has no FFI of its own: the seam it replaces is the Rust function
 *
 *     pub fn render(target: &mut RenderTarget, scene: &Scene)          lib/core/render.rs:61
 *
 * -- precisely its closure + rayon loop, render.rs:74-126 -- together with the data the closure
 * reaches through the traits of lib/core/traits.rs:13-30 and BoundsCollection (lib/core/bvh.rs:18-28).
 * A `Scene` is lowered to the plain-old-data structs below (INTEGRATION.md shows the Rust `extern "C"`
 * block and the lowering a maintainer would add); the entry points then do what render() does:
 *
 *     rtr_upload_scene   <-  scene.build_bvh() (render.rs:67, scene.rs:112) + Mesh::build_bvh (mesh.rs:119):
 *                            the reference's median-split builder re-implemented on the host, flattened
 *                            to a linear node array + SoA leaf records in HBM
 *     rtr_render*        <-  the per-pixel closure calculate_pixel (render.rs:75-111), trace_ray
 *                            (render.rs:32-59), world.hit (bvh.rs:146-199 | scene.rs:141-157),
 *                            Material::scatter (materials.rs), Background::sample (backgrounds.rs),
 *                            clamp / mean / sqrt / RGBA8 (render.rs:95-110, color.rs:34-41)
 *
 * Conventions (reference conventions in brackets):
 *   - every function returns RTR_OK or a negative rtr_status; nothing unwinds across the boundary
 *     [the reference panics: scene.rs:75, mesh.rs:65, bvh.rs:99, render.rs:95];
 *   - the caller owns every buffer; inputs are copied during rtr_upload_scene and no host pointer is
 *     retained; host outputs are complete when the call returns [render() is synchronous, render.rs:61];
 *   - calls on one context serialise on an internal mutex (one frame at a time, like the reference's blocking render());
 *     distinct contexts are independent;
 *   - pixel layout is the reference's: row-major, y = 0 is the top row, RGBA8 with A = 255
 *     (render.rs:22-29,110);
 *   - random numbers: the reference uses an OS-seeded thread_rng() and is not reproducible; here every
 *     (seed, pixel index, sample index) owns a counter-based stream, so a frame is a pure function of
 *     (scene, camera, settings, size, seed) for any number of GPUs.
 *
 * There is no CPU fallback: every compute entry point fails with RTR_ERR_NO_DEVICE when no gfx950
 * device is usable.
 */
#ifndef RTR_HIP_H
#define RTR_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* 2: volumes, rtr_set_option.  3: rtr_selftest_div, options "leaf_pairs".  4: mesh-bounded volumes.
 * 5: rtr_create_multi -- one context over several GPUs --, rtr_probe_hits, rtr_camera_ray_origin, traversal counters in rtr_stats. */
#define RTR_ABI_VERSION 5

typedef enum rtr_status {
    RTR_OK = 0,
    RTR_ERR_INVALID_ARGUMENT = -1,
    RTR_ERR_NO_DEVICE = -2,        /* no HIP device / context created host-only */
    RTR_ERR_HIP = -3,              /* a HIP runtime call failed; see rtr_last_error */
    RTR_ERR_BAD_INDEX = -4,        /* material / vertex / object index out of bounds (scene.rs:106-110 panics) */
    RTR_ERR_NAN_CENTROID = -5,     /* partial_cmp().unwrap() in the BVH builder would panic (bvh.rs:99,109,119) */
    RTR_ERR_BAD_SETTINGS = -6,     /* clamp_indirect NaN/negative (f32::clamp panics, render.rs:95), empty target */
    RTR_ERR_NO_SCENE = -7,         /* render before rtr_upload_scene */
    RTR_ERR_UNSUPPORTED = -8,      /* scene feature outside the GPU path (Volumes nested more than one level deep, LDS budget) */
    RTR_ERR_OUT_OF_MEMORY = -9
} rtr_status;

/* lib/materials.rs, lib/volume.rs:75-88 */
typedef enum rtr_material_kind {
    RTR_MAT_LAMBERTIAN = 0, /* rgb = albedo                     materials.rs:18-42  */
    RTR_MAT_METAL = 1,      /* rgb = albedo, param = fuzz       materials.rs:44-71  */
    RTR_MAT_EMISSION = 2,   /* rgb = color*strength             materials.rs:73-91  */
    RTR_MAT_DIELECTRIC = 3, /* param = ir                       materials.rs:93-139 */
    RTR_MAT_MIX = 4,        /* param = factor, first/second = indices into the same materials array, both
                               greater than the mix's own index (nested boxes flattened)  materials.rs:141-157 */
    RTR_MAT_ISOTROPIC = 5   /* rgb = color                      volume.rs:75-88 */
} rtr_material_kind;

typedef enum rtr_object_kind { RTR_OBJ_SPHERE = 0, RTR_OBJ_MESH = 1, RTR_OBJ_VOLUME = 2 } rtr_object_kind;
typedef enum rtr_background_kind { RTR_BG_UNIFORM = 0, RTR_BG_GRADIENT = 1, RTR_BG_SKYMAP = 2 } rtr_background_kind;
/* RenderMode, lib/core/scene.rs:14-20 */
typedef enum rtr_render_mode { RTR_MODE_FULL = 0, RTR_MODE_CLAY = 1, RTR_MODE_RANDOM = 2, RTR_MODE_NORMAL = 3 } rtr_render_mode;

/* Sphere, lib/objects.rs:15-20 */
typedef struct rtr_sphere {
    float center[3];
    float radius;
    uint32_t material;
} rtr_sphere;

/* Mesh::from_buffers inputs, lib/core/mesh.rs:78-82: (position, normal) vertex buffer + u32 triangle list */
typedef struct rtr_mesh {
    const float *positions; /* n_verts * 3 */
    const float *normals;   /* n_verts * 3 */
    const uint32_t *indices; /* n_tris * 3 */
    uint32_t n_verts;
    uint32_t n_tris;
    uint32_t material;
    uint32_t reserved;
} rtr_mesh;

/* Volume, lib/volume.rs:19-32: a constant-density medium inside a boundary object (`boundary: Box<dyn Object>`).
 * boundary_kind = RTR_OBJ_SPHERE: `boundary` is the sphere.  boundary_kind = RTR_OBJ_MESH (ABI revision 4):
 * meshes[boundary_mesh] is the boundary; Volume::hit then makes its two boundary.hit calls (volume.rs:35-36) through
 * that mesh's own tree with the ranges [-inf, +inf] and [t1 + tmin, +inf] -- Triangle::hit takes the lower bound as its
 * determinant epsilon too (mesh.rs:166,190), which is reproduced.  boundary_kind = RTR_OBJ_VOLUME (ABI revision 5):
 * volumes[boundary_mesh] -- the same index field -- is the boundary, itself a Volume over a sphere or a mesh (one level of
 * nesting; deeper chains are RTR_ERR_UNSUPPORTED): the two boundary.hit calls are then Volume::hit again, each with its
 * own random draw, in call order.  The boundary's material is the medium's (normally Isotropic).  A boundary sphere, mesh
 * or Volume is NOT listed in objects[] itself. */
typedef struct rtr_volume {
    rtr_sphere boundary;
    float neg_inv_density; /* -1/density, as Volume::new stores it (volume.rs:29) */
    uint32_t boundary_kind; /* RTR_OBJ_SPHERE (0) | RTR_OBJ_MESH (1) | RTR_OBJ_VOLUME (2) */
    uint32_t boundary_mesh; /* index into meshes[] (boundary_kind = RTR_OBJ_MESH) or into volumes[] (RTR_OBJ_VOLUME) */
} rtr_volume;

typedef struct rtr_material {
    uint32_t kind; /* rtr_material_kind */
    float rgb[3];
    float param;
    uint32_t first;
    uint32_t second;
} rtr_material;

/* Scene.objects is heterogeneous and ordered (scene.rs:37); order decides BVH ties and the linear scan */
typedef struct rtr_object {
    uint32_t kind;  /* rtr_object_kind */
    uint32_t index; /* into spheres[], meshes[] or volumes[] */
} rtr_object;

/* lib/backgrounds.rs */
typedef struct rtr_background {
    uint32_t kind;      /* rtr_background_kind */
    float a[3];         /* uniform colour | gradient top */
    float b[3];         /* gradient bottom */
    const float *image; /* skymap: width*height RGB float triples, row 0 = top scanline (backgrounds.rs:68) */
    uint32_t width;
    uint32_t height;
} rtr_background;

typedef struct rtr_scene_desc {
    const rtr_object *objects;
    uint32_t n_objects;
    const rtr_sphere *spheres;
    uint32_t n_spheres;
    const rtr_mesh *meshes;
    uint32_t n_meshes;
    const rtr_material *materials;
    uint32_t n_materials;
    rtr_background background;
    const rtr_volume *volumes; /* appended in ABI revision 2 */
    uint32_t n_volumes;
} rtr_scene_desc;

/* Camera, lib/core/camera.rs:27-34 */
typedef struct rtr_camera {
    float lookfrom[3];
    float lookat[3];
    float vertical[3];
    float vertical_fov; /* degrees */
    float aspect_ratio; /* NOT derived from the target size (camera.rs:41) */
} rtr_camera;

/* RenderSettings, lib/core/scene.rs:22-30 (field for field) */
typedef struct rtr_settings {
    uint32_t samples_per_pixel;
    uint32_t max_ray_depth;         /* u8 in the reference: values above 255 are rejected */
    float clamp_indirect;
    uint32_t enable_multithreading; /* accepted for parity; the GPU path is always parallel */
    uint32_t enable_bvh_tree;
    uint32_t mode;                  /* rtr_render_mode */
} rtr_settings;

typedef struct rtr_scene_info {
    uint32_t n_objects, n_spheres, n_meshes, n_triangles, n_materials;
    uint32_t n_leaves;      /* leaves of the flattened two-level tree (spheres + triangles) */
    uint32_t n_nodes;       /* inner (Split) nodes */
    uint32_t tree_height;   /* longest root-to-leaf path in Split nodes: the traversal stack bound */
    uint32_t spheres_only;  /* 1 when every object is a sphere */
    uint64_t device_bytes;  /* HBM resident bytes for this scene */
} rtr_scene_info;

typedef struct rtr_stats {
    uint64_t primary_samples; /* W*H*spp of the last render (this shard's share) */
    uint64_t segments;        /* ray segments = world.hit calls (render.rs:38) counted on the device */
    double kernel_ms;         /* HIP-event time of the last render kernel on its stream */
    double resolve_ms;        /* HIP-event time of the last resolve kernel (0 when fused) */
    double upload_ms;         /* host build + flatten + copy of the last rtr_upload_scene (wall) */
    uint32_t launches;        /* render kernels launched by this context so far */
    uint32_t vgprs, lds_bytes, block_threads; /* launch configuration of the last render kernel */
    /* ABI revision 5 -- traversal work of the last render, counted on the device when option "traversal_counters" is on and
     * the launch walked the tree (traversal_counted = 1; else all zero).  node_steps: BvhNode::Split visits -- a node step
     * tests the boxes of BOTH children (bvh.rs:160-166 / aabb.rs:81-137), so the reference's count of AABB::hit calls under
     * its unculled both-children traversal compares with 2 * node_steps; leaf_tests: Sphere::hit / Triangle::hit /
     * Volume::hit calls reached through the tree; sky_lookups: SkyMap::sample calls (backgrounds.rs:84-92). */
    uint64_t node_steps, leaf_tests, sky_lookups;
    uint32_t traversal_counted, reserved;
} rtr_stats;

typedef struct rtr_context rtr_context;

/* device >= 0: bind that HIP device.  device == -1: host-only context (scene lowering, BVH build and the
 * introspection calls work; every compute call returns RTR_ERR_NO_DEVICE). */
int rtr_create(int device, rtr_context **out);
/* One context over the n_devices GPUs device_ids[] of this node (what render()'s rayon fan-out,
 * lib/core/render.rs:114-119, becomes on a node of GPUs).  rtr_upload_scene replicates the scene on every GPU; rtr_render,
 * rtr_render_rgba8, rtr_render_f32 and rtr_render_frame_device divide each frame among the GPUs -- by samples (default: every
 * GPU integrates all pixels for a contiguous share of the samples; the random streams are keyed by the absolute sample
 * index, so the frame differs from the single-GPU frame only in the order of the float additions across GPUs: ~1e-7
 * relative, RGBA8 within 1) or by 8x8 tiles (option "multi_split" = "tiles": bit-identical to one GPU) --, sum the GPUs'
 * partial sums onto device_ids[0] with one ncclReduce per frame (RCCL over xGMI, bound at run time) and resolve there
 * (mean, sqrt gamma, RGBA8: render.rs:103-110).  One host thread and two streams per GPU; partial sums are double
 * buffered, so with rtr_render_frame_device -- asynchronous; its d_rgba8 lives on device_ids[0], its stream argument is
 * ignored -- frame i's reduce and resolve overlap frame i+1's integrate.  rtr_get_stats / rtr_segment_counts add the GPUs'
 * segments, the *_times_ms calls report the slowest GPU.  The per-shard entry points (rtr_render_tiles_device,
 * rtr_render_samples_device) return RTR_ERR_UNSUPPORTED on such a context.
 * device_ids must be distinct; all -1: a host-only context; all the same GPU: a test configuration without RCCL (the
 * partial sums are added by a kernel).  n_devices = 1 is a single-GPU context that goes through the same reduce. */
int rtr_create_multi(const int *device_ids, int n_devices, rtr_context **out);
/* The samples [*begin, *begin + *count) of a frame of samples_per_pixel samples that GPU `rank` of `n_ranks` integrates
 * under the sample split: contiguous, the first samples_per_pixel % n_ranks ranks take one more.  Pure arithmetic. */
int rtr_sample_range(uint32_t rank, uint32_t n_ranks, uint32_t samples_per_pixel, uint32_t *begin, uint32_t *count);
void rtr_destroy(rtr_context *ctx);
/* message for the last failing call on ctx (ctx == NULL: last failing rtr_create on this thread) */
const char *rtr_last_error(const rtr_context *ctx);
int rtr_abi_version(void);

/* Context options (strings; applied at the next rtr_upload_scene):
 *   "mesh_tree" = "reference" (default): every mesh subtree is the reference's median-split tree (mesh.rs:119-121),
 *                 the parity configuration;
 *               = "sah": mesh subtrees rebuilt with a binned surface-area heuristic over the same triangle boxes
 *                 (SURVEY section 8f N1).  Closest hits are the same and exact-t ties still resolve by the reference's
 *                 depth-first leaf rank; only ulp-level box-boundary cases can differ.  About half the traversal work
 *                 on assets/house.obj.
 *               = "lbvh": mesh subtrees are built ON THE DEVICE: Morton codes of the triangle boxes'
 *                 centroids, a radix sort, Karras' parallel radix tree, one bottom-up pass for the boxes -- instead of the
 *                 host's O(n log^2 n) median-split build; for large or changing meshes.  Same closest hits; exact-t ties
 *                 resolve by triangle index (the reference's depth-first rank is not computed); box-boundary grazes can
 *                 differ as with "sah".  Needs a device-bound context.  Meshes that bound a Volume keep the reference tree.
 *   "scene_tree" = "reference" (default) | "sah": the same choice for the scene-level tree over the objects; honoured
 *                 only for scenes without Volume objects whose object boxes all have positive extent.
 *   "leaf_pairs" = "on" (default) | "off": a Split node whose two children are leaves of one type with consecutive
 *                 ranks is stored inside its parent (same box test, same two leaf tests in the same order, one traversal
 *                 step fewer).  Decisions are identical either way -- frames are bit-identical; "off" exists for A/B
 *                 measurements.
 *   "tree_kernel" = "auto" (default) | "stream" | "samples" | "resumable" | "wave" (applied at the next render): which kernel
 *                 walks the tree when enable_bvh_tree is set.  "wave": a wave returns from world.hit when its slowest
 *                 lane has finished.  "resumable": the walk is a per-lane state machine inside the main loop, so lanes
 *                 that are done shade and start their next ray while the others keep walking.  "samples": the same
 *                 walk, but a lane takes (sample, pixel) pairs instead of owning one pixel -- the launch is one workgroup
 *                 per resident wave, and a wave claims blocks of (tile, "block_samples" samples) from a device counter; each
 *                 sample's clamped radiance is written to device memory (16 bytes per sample, in batches of at most 16 GiB
 *                 and fewer than 2^32 (sample, pixel) pairs)
 *                 and a second kernel adds every pixel's samples in index order.  "stream": the streaming wavefront --
 *                 the paths of a batch of samples live in device memory (65 + 4 * max_ray_depth bytes each) and every
 *                 stage is its own kernel: generate, then per recursion depth an intersect kernel whose lanes take the
 *                 next waiting path the moment their walk ends and a shade kernel over full waves, then the same
 *                 in-order sum as "samples".  Every lane-level operation and every addition is the same in all of
 *                 them: bit-identical frames.  "auto": trees of 32 leaves or more
 *                 take "samples" (inside tile-sharded launches, where it does not apply: "resumable" for scenes with
 *                 meshes, else "wave"); smaller trees "wave". */
/*   "traversal_counters" = "off" (default) | "on" (applied at the next render): tree launches take the counting instantiation
 *                 of the "samples" kernel (the same walk; its lanes also add up their node steps, leaf tests and SkyMap lookups)
 *                 and rtr_get_stats reports them -- how much of the reference's unculled traversal the ordered, culling walk
 *                 skips.  Costs a few per cent; the frame is unchanged.
 *   "multi_split" = "samples" (default) | "tiles" (contexts of rtr_create_multi; applied at the next render): see there. */
/*   "samples_batch" = "0" (default: what half of the free device memory holds, at most 16 GiB of radiance and 64
 *                 samples for "stream") | N: samples per batch of the "samples" / "stream" kernels' buffers; a pixel's
 *                 running sum is carried from batch to batch, so any N gives the same frame.  A batch whose allocation
 *                 fails is halved and retried.
 *   "stream_flush" = "8" (default) | 1 .. 64: finished walks a wave of the "stream" intersect kernel collects before it
 *                 stores their hits and hands the lanes new paths (tuning; any value gives the same frame). */
/*   "sample_chunks" = "auto" (default) | "1" .. "4096" (applied at the next render): pieces every pixel's sample range
 *                 is cut into inside one launch.  Each piece continues the running sums of the one before it in the
 *                 same order, so the frame is bit-identical for every value; more pieces shorten the tail of a launch
 *                 that has few tiles per resident wave and cost a workgroup prologue and a ragged end each.  "auto": the
 *                 sphere world takes three pieces in the proportions 24:7:1 from 96 samples on and two (7:1) from 28
 *                 (1080p, 256 samples: 13.0 ms against 14.7 in one piece); an explicit count is cut with linearly falling
 *                 lengths.  The hand-over between pieces is made of device-scope atomic loads and stores only.
 *   "debug_chunk_lengths" = "a:b:..." (1 .. 8 positive integers): measuring aid, the pieces' relative lengths (sets the
 *                 count too); "debug_chunk_taper" = "auto" | 0 .. 7: how steeply an explicit count's lengths fall.
 *   "chunk_timeout_ms" = "10000" (default) | milliseconds: how long a piece waits for the piece before it.  HIP does
 *                 not promise dispatch order, so the wait is bounded; a piece that gives up sets the context's device
 *                 fault word and the next synchronising call (rtr_render*, rtr_get_stats, rtr_kernel_times_ms,
 *                 rtr_segment_counts) returns RTR_ERR_HIP once: the frames rendered since the previous such call
 *                 are invalid.
 *   "sky_filter" = "nearest" (default: SkyMap::sample, backgrounds.rs:84-92) | "bilinear": interpolate the four texels
 *                 around the direction (longitude wraps, latitude clamps).  An extension, not reference behaviour.
 *   "debug_fault" = "off" (default) | "on": testing aid; launches set the fault word themselves.
 *   "block_samples" = "8" (default) | 1 .. 1024: the samples kernel runs one workgroup per resident wave; a wave whose lanes
 *               ask for work claims the next block -- one 8x8 tile, this many consecutive samples -- from a counter in
 *               device memory.  Any value gives the same frame.
 *   "stack_entries" = "auto" (default) | "wide": the samples kernel keeps a traversal-stack entry in one 32-bit word (4 flag
 *               bits, 16-bit index, 12 bits of entry distance) for trees of at most 65 536 nodes and leaves, else in a
 *               32-bit ref plus a 16-bit distance; "wide" forces the second form (tests).  The distance is a lower bound used
 *               only to skip boxes the best hit has overtaken: the frame does not depend on the form, the LDS footprint
 *               (and with it the occupancy: LDS is granted in 1280-byte steps) does.
 *   "debug_lds_pad" = "0" (default) | bytes up to 65536: measuring aid, extra dynamic LDS per workgroup.
 *   "debug_alloc_limit" = "0" (default: none) | bytes: testing aid; scratch allocations above that size fail as if the
 *                 device were out of memory (exercises the batch-halving retry of the "samples" / "stream" kernels).
 * A scratch buffer that has to grow is freed and reallocated inside the call that needs it, which synchronises the
 * device once; sizes are remembered, so steady-state frames of one size never reallocate.
 * Scratch shared by the launches of ONE context: the running sums and counters of sample chunks, the radiance buffer
 * and running sums of the "samples" tree kernel, the event / counter ring.  Issue a context's launches on one stream
 * (or order them with events); use one context per stream for concurrent frames. */
int rtr_set_option(rtr_context *ctx, const char *name, const char *value);
/* Lower + build + flatten + copy to HBM.  Replaces any previous scene of the context. */
int rtr_upload_scene(rtr_context *ctx, const rtr_scene_desc *scene);
int rtr_scene_info_get(const rtr_context *ctx, rtr_scene_info *out);
/* Leaves of the flattened tree in the reference's depth-first order (left before right, bvh.rs:162-163).
 * out[i] = (object index << 32) | triangle handle (0 for spheres).  Returns the leaf count (<= cap used). */
int64_t rtr_bvh_leaf_order(const rtr_context *ctx, uint64_t *out, uint64_t cap);

/* render(): the whole frame, host outputs.  rgba8: width*height*4 bytes or NULL.  mean_rgb: width*height*3
 * floats, colour/spp BEFORE the sqrt gamma (render.rs:103-107), or NULL. */
int rtr_render(rtr_context *ctx, const rtr_camera *camera, const rtr_settings *settings, uint32_t width,
               uint32_t height, uint64_t seed, uint8_t *rgba8, float *mean_rgb);
int rtr_render_rgba8(rtr_context *ctx, const rtr_camera *camera, const rtr_settings *settings, uint32_t width,
                     uint32_t height, uint64_t seed, uint8_t *rgba8);
int rtr_render_f32(rtr_context *ctx, const rtr_camera *camera, const rtr_settings *settings, uint32_t width,
                   uint32_t height, uint64_t seed, float *mean_rgb);

/* --- device-resident entry points (multi-GPU sharding, benchmarking); all asynchronous on `stream`
 *     (a hipStream_t; NULL = the context's own stream) --------------------------------------------- */

/* Fused single-GPU frame: integrate + clamp + mean + sqrt + RGBA8, output in HBM.
 * d_rgba8: device pointer, width*height*4 bytes. */
int rtr_render_frame_device(rtr_context *ctx, const rtr_camera *camera, const rtr_settings *settings, uint32_t width,
                            uint32_t height, uint64_t seed, uint8_t *d_rgba8, void *stream);
/* Sharded integrate: 8x8-pixel tiles t with t % n_shards == shard are rendered; for each of their pixels the
 * sum over samples of the clamped radiance (render.rs:95-99) is STORED to d_sum_rgb[pixel*3 .. +2] (three floats per
 * pixel, packed: ABI revision 5; revisions up to 4 stored a padded float4).
 * Pixels of other shards are not touched: zero the buffer first, then a sum over shards (RCCL reduce) is the
 * full frame, bit-identical to the single-GPU sum because x + 0 = x. */
int rtr_render_tiles_device(rtr_context *ctx, const rtr_camera *camera, const rtr_settings *settings, uint32_t width,
                            uint32_t height, uint64_t seed, uint32_t shard, uint32_t n_shards, float *d_sum_rgb,
                            void *stream);
/* Sample-sharded integrate: EVERY pixel, samples [sample_begin, sample_begin + sample_count) of the frame's
 * samples_per_pixel; the in-order sum of their clamped radiance is STORED to d_sum_rgb[pixel*3 .. +2].  The random
 * streams are keyed by the absolute sample index, so the shards of a frame draw exactly the single-launch numbers;
 * only the order of the float additions across shards differs (a sum of partial sums: ~1e-7 relative).
 * This is the multi-GPU decomposition bench.py uses: every GPU keeps a full grid of short waves (perfect balance). */
int rtr_render_samples_device(rtr_context *ctx, const rtr_camera *camera, const rtr_settings *settings, uint32_t width,
                              uint32_t height, uint64_t seed, uint32_t sample_begin, uint32_t sample_count,
                              float *d_sum_rgb, void *stream);
/* mean = sum/spp, sqrt gamma, RGBA8 (render.rs:103-110).  d_rgba8 and/or d_mean_rgb may be NULL. */
int rtr_resolve_device(rtr_context *ctx, const float *d_sum_rgb, uint32_t samples_per_pixel, uint32_t width,
                       uint32_t height, uint8_t *d_rgba8, float *d_mean_rgb, void *stream);

/* Waits for the context's outstanding work, then reports counters and HIP-event timings.  Like rtr_kernel_times_ms
 * and rtr_segment_counts it returns RTR_ERR_HIP (once) when a launch since the last synchronising call set the device
 * fault word (option "chunk_timeout_ms"): the asynchronous entry points cannot report that themselves. */
int rtr_get_stats(rtr_context *ctx, rtr_stats *out);
/* HIP-event durations (ms) of the most recent render kernels, oldest first; returns how many were written
 * (at most cap, at most 64).  Waits for outstanding work. */
int rtr_kernel_times_ms(rtr_context *ctx, float *out, uint32_t cap);
/* The same launches: HIP-event time from the start of each launch to the end of its (last) integrate kernel, i.e.
 * without the k_sum_samples pass of the "samples" tree kernel.  bench.py prices the dominant kernel with it. */
int rtr_integrate_times_ms(rtr_context *ctx, float *out, uint32_t cap);
/* Name of the integrate kernel the most recent render launched (as rocprofv3 lists it, with a gloss).  The pointer is to a
 * per-thread copy: valid until the calling thread's next rtr_kernel_name call. */
const char *rtr_kernel_name(const rtr_context *ctx);
/* Device self-test: the kernels' correctly-rounded square root (csrc/rtr_kernels.hip rt_sqrt) against the
 * compiler's sqrtf() over all 2^32 f32 bit patterns; *mismatches must come back 0. */
int rtr_selftest_sqrt(rtr_context *ctx, uint64_t *mismatches);
/* Device self-test: the kernels' shared-denominator division (csrc/rtr_kernels.hip shared_div: camera u,v and sphere
 * normals) against the compiler's correctly rounded n / denominator, for every f32 bit pattern n in
 * [numerator_lo_bits, numerator_hi_bits]; *mismatches must come back 0 for 2^-20 <= |denominator| <= 2^40 and
 * numerators that are +0 or have 2^-40 <= |n| <= 2^60 (a denominator outside its range counts as one mismatch). */
int rtr_selftest_div(rtr_context *ctx, float denominator, uint32_t numerator_lo_bits, uint32_t numerator_hi_bits,
                     uint64_t *mismatches);
/* Device-counted ray segments of the same launches, in the same order. */
int rtr_segment_counts(rtr_context *ctx, uint64_t *out, uint32_t cap);

/* --- introspection of the path's closed forms (known-answer tests run them on the product, not on a restatement) --- */

/* HitRecord, lib/utils/ray.rs:29-54, as world.hit returns it to trace_ray (render.rs:38) */
typedef struct rtr_hit {
    uint32_t hit;        /* 0: None -- the other fields are then unspecified */
    float t;
    float point[3];
    float normal[3];     /* face-forwarded (ray.rs:45-49) */
    uint32_t front_face;
    uint32_t material;   /* MaterialHandle */
    uint32_t reserved[2];
} rtr_hit;
/* world.hit(ray, 0.00001, +inf) (render.rs:38) of the uploaded scene for n_rays caller-supplied rays (origins, directions:
 * n_rays * 3 floats, host memory; directions need not be normalised, as in the reference), on the device, through the same
 * code the render kernels inline: the BvhTree walk (bvh.rs:43-50,146-199) when enable_bvh_tree, else Scene::hit's linear scan
 * (scene.rs:141-157).  A Volume's random draw (volume.rs:57) comes from the stream (seed, pixel = ray index, sample 0). */
int rtr_probe_hits(rtr_context *ctx, uint32_t enable_bvh_tree, uint64_t seed, uint32_t n_rays, const float *origins,
                   const float *directions, rtr_hit *out);
/* Camera::ray_origin (lib/core/camera.rs:37-56): out = origin[3], horizontal[3], vertical[3], lower_left_corner[3] -- the
 * values every launch derives from rtr_camera.  Host arithmetic (the reference's tan is host code too); needs no context. */
int rtr_camera_ray_origin(const rtr_camera *camera, float out[12]);

#ifdef __cplusplus
}
#endif
#endif /* RTR_HIP_H */
