// rtr_device.h -- plain-old-data shared by the host flattener and the gfx950 kernels.
//
// HBM layout of an uploaded scene (DESIGN.md "Data layout"):
//   nodes[]      64 B per Split node of the reference's tree (lib/core/bvh.rs:52-56), preorder.  A node
//                carries the boxes of its two CHILDREN, so one 64-B gather decides both descents; the
//                box of the root lives in the launch parameters.  Leaves (bvh.rs `Object(u32)`) have no
//                box in the reference and none here.
//   leaf_hot[]   48 B per leaf, indexed by the leaf's rank L in the reference's depth-first order
//                (left before right, bvh.rs:162-163).  Sphere: centre, radius^2 (one 16-byte load per test), radius.
//                Triangle: v0, e1 = v1-v0, e2 = v2-v0 (mesh.rs:160-161, hoisted: pure functions).
//                Volume: boundary centre, radius, radius^2, -1/density (volume.rs:19-32).
//   leaf_cold[]  64 B per leaf, read once per segment for the winning leaf only: vertex normals,
//                face normal (mesh.rs:86-90), material handle.
//   lin_spheres[] 32 B per sphere in Scene.objects order: the linear world of scene.rs:141-157 when
//                every object is a sphere (scenes/benchmark.json); read with wave-uniform indices.
//   objects[]    32 B per object in Scene.objects order: root ref + root box, for the linear world
//                with meshes.
//   materials[]  32 B per material.
//   sky[]        16 B per texel (RGB + pad) for SkyMap (backgrounds.rs:51-56).
#pragma once
#include <stdint.h>

namespace rtr {

constexpr uint32_t REF_NONE = 0xFFFFFFFFu;  // BvhNode::None
constexpr uint32_t REF_LEAF = 0x80000000u;  // BvhNode::Object
constexpr uint32_t REF_TRI = 0x40000000u;   // leaf is a triangle
constexpr uint32_t REF_VOL = 0x20000000u;   // leaf is a Volume (lib/volume.rs); neither bit: sphere
constexpr uint32_t REF_PAIR = 0x10000000u;  // leaf ref only: a collapsed Split of the two leaves L and L+1 (same
                                            // type), gated by the Split's box like an inner node
constexpr uint32_t REF_INDEX = 0x0FFFFFFFu; // leaf rank L, or node index

constexpr uint32_t NODE_LVOL = 1u; // the left / right subtree holds a Volume leaf: Volume::hit draws a random
constexpr uint32_t NODE_RVOL = 2u; // number on every visit, so such subtrees are never culled nor reordered

struct alignas(16) DevNode {
    float lmin[3], lmax[3];
    float rmin[3], rmax[3];
    uint32_t lref, rref;
    uint32_t flags, pad1;
};
static_assert(sizeof(DevNode) == 64, "node is one 64-byte gather");

struct alignas(16) DevLeafHot {
    float f[12];
};
static_assert(sizeof(DevLeafHot) == 48, "hot leaf record");

struct alignas(16) DevLeafCold {
    float n0[3]; uint32_t material;
    float n1[3]; uint32_t object;
    float n2[3]; uint32_t handle;
    float fn[3]; uint32_t pad;
};
static_assert(sizeof(DevLeafCold) == 64, "cold leaf record");

struct alignas(16) DevLinSphere {
    float center[3]; float radius;
    float radius2; uint32_t material; uint32_t pad0, pad1;
};
static_assert(sizeof(DevLinSphere) == 32, "linear-world sphere");

struct alignas(16) DevObject {
    uint32_t ref;
    float box[6]; // root box when ref is a Split node
    uint32_t pad;
};
static_assert(sizeof(DevObject) == 32, "object record");

struct alignas(16) DevMaterial {
    float rgb[3]; float param;
    uint32_t kind, first, second, pad; // Mix: first / second = child handles.  Dielectric: f32 bits of 1/ir, r0^2 (front), r0^2 (back)
};
static_assert(sizeof(DevMaterial) == 32, "material record");

enum WorldKind : uint32_t {
    WORLD_LINEAR_SPHERES = 0, // scene.rs:141-157, every object a sphere
    WORLD_BVH = 1,            // bvh.rs:43-50 over the two-level tree
    WORLD_LINEAR_GENERIC = 2  // scene.rs:141-157 with meshes (each mesh traverses its own tree, mesh.rs:124-127)
};

// Launch parameters: one kernel argument block (scalar registers / kernarg segment).
struct RenderParams {
    // RayOrigin, camera.rs:11-17
    float cam_origin[3], cam_horizontal[3], cam_vertical[3], cam_llc[3];
    float width_m1, height_m1; // (W-1) as f32, (H-1) as f32  (render.rs:88-89)
    uint32_t width, height;
    uint32_t spp, max_depth, mode;   // spp: the frame's samples per pixel (the divisor of the fused resolve)
    uint32_t sample_begin, sample_count; // the samples this launch integrates (whole frame: 0, spp)
    float clamp_indirect;
    uint32_t seed_key0, seed_key1; // two 32-bit hashes of the 64-bit seed (rtr_api.cpp seed_keys; the oracle's or_rng_seed)
    // world
    uint32_t world, n_objects;
    uint32_t root_ref;
    float root_box[6];
    uint32_t stack_cap; // entries per lane
    uint32_t chain_bits; // != 0: a path's attenuation chain is kept in one register, this many bits per material id
    uint32_t chain_lds;  // words per lane of the LDS chain region: max_depth ids (3 * max_depth in the sphere world, which keeps
                         // the attenuations themselves), or 0 with chain_bits
    uint32_t sphere_regs;   // linear sphere world: spheres held in scalar registers (0 = generic loop)
    uint32_t plain_sphere_world; // ... in RenderMode::Full over a UniformBackground, no MixMaterial / Isotropic, attenuations in the chain:
                            // the instantiation without those wave-uniform tests (shade_segment<PLAIN>)
    uint32_t lds_materials; // number of material records staged in LDS (0 = read from HBM/L1)
    uint32_t has_meshes;    // the scene holds triangles: tree kernels instantiated with the unconditional slab tests
    uint32_t sample_parallel; // WORLD_BVH: k_render_bvh_sp + k_sum_samples
    uint32_t resumable_walk; // WORLD_BVH: k_render_bvh (per-lane walk state machine) instead of k_render
    uint32_t has_volumes;   // 0: no Volume objects; 1: sphere-bounded ones (kernels instantiated with the visit-order rules);
                            // 2: mesh-bounded ones too (those kernels plus the nested mesh walk of Volume::hit)
    uint32_t *work_counter; // the next unclaimed ticket of the launch (k_render, k_render_bvh: (chunk, tile), chunk-major, one per
                            // workgroup; sample-parallel kernel: (tile, group of samples_per_block samples), claimed by resident waves)
    uint32_t work_blocks;   // tickets of this launch: tiles_launch * chunks | tiles_launch * ceil(sample_count / samples_per_block)
    uint32_t samples_per_block;
    uint32_t stack_packed; // 1: a traversal-stack entry is one word -- 4 flag bits, a 16-bit index, 12 bits of entry distance
    uint32_t stack_pad;    //    (trees of at most 65 536 nodes and leaves); 0: a 32-bit ref plus a 16-bit distance
    const DevNode *nodes;
    const DevLeafHot *leaf_hot;
    const DevLeafCold *leaf_cold;
    const DevLinSphere *lin_spheres;
    const DevObject *objects;
    const DevMaterial *materials;
    uint32_t n_materials;
    // background
    uint32_t bg_kind;
    float bg_a[3], bg_b[3];
    const float *sky; // float4 texels
    uint32_t sky_w, sky_h;
    uint32_t sky_bilinear; // option "sky_filter" = "bilinear": an extension, the reference takes the nearest texel
    // tile sharding
    uint32_t tiles_x, tiles_total, shard, n_shards;
    uint32_t tiles_launch;  // tiles this launch renders (shard, shard + n_shards, ...): grid = tiles_launch * chunks
    uint32_t chunks;        // pieces every pixel's sample range is cut into (>= 1); tickets are chunk-major
    uint32_t chunk_taper;   // 0: equal pieces; k in 1..7: lengths fall linearly from (8+k)/8 to (8-k)/8 of the mean
    uint32_t chunk_table_on; // 1: chunks <= 8 and chunk c covers samples [chunk_first[c], chunk_first[c + 1]) of the launch's range
    uint32_t chunk_first[9]; //    (filled by the host: no 64-bit divisions in a workgroup's prologue); 0: chunk_boundary(), any count
    float *carry_rgba;      // float4 running sums handed from chunk to chunk (chunks > 1)
    float *samples;         // k_render_bvh_sp: float4 [sample - sample_begin][pixel], one clamped radiance per sample
    uint32_t *chunk_flags;  // per launch-local tile: chunks completed (chunks > 1)
    unsigned long long chunk_timeout_ticks; // bound of a chunk's wait for its predecessor, in s_memrealtime ticks (100 MHz)
    unsigned long long *fault; // the context's sticky fault word: bit 0 = a chunk gave up waiting, bit 1 = "debug_fault"
    // streaming wavefront path (tree worlds, option tree_kernel = "stream"): one record per path of a batch in HBM
    float *wf_ray;        // 2 float4 per path: (o.xyz, d.x), (d.y, d.z, 0, 0)
    float *wf_hit;        // float4 per path: t, u, v, leaf ref (bits); written by k_wf_intersect, read by k_wf_shade
    uint32_t *wf_state;   // uint4 per path: rng state, rng inc, chain length, slot of the path's radiance in `samples`
    uint32_t *wf_chain;   // [max_depth][wf_paths] material ids of the attenuation chain
    uint8_t *wf_stage;    // per path: the depth at which it waits for intersect + shade; 255 = finished, or no such pixel
    uint32_t wf_paths;    // paths of this batch: tiles * 64 * samples
    uint32_t wf_blocks;   // ceil(wf_paths / 64): the unit waves scan
    uint32_t wf_depth;    // the depth the launched kernel serves (trace_ray's `depth` argument, render.rs:32)
    uint32_t wf_flush;    // k_wf_intersect: finished walks a wave collects before it stores them and refills the lanes
    // outputs
    float *sum_rgb;              // three floats per pixel, packed (sharded path) or nullptr
    uint8_t *rgba8;              // fused path or nullptr
    float *mean_rgb;             // fused path, optional
    unsigned long long *counter; // segments
    unsigned long long *walk_counters; // option "traversal_counters": node steps, leaf tests, SkyMap lookups of the launch (3 words), or nullptr
};

} // namespace rtr
