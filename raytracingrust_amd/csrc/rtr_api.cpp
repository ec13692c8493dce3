// rtr_api.cpp -- implementation of the C ABI declared in include/rtr_hip.h (host side, HIP runtime).
//
// A context owns one HIP device, one stream, the scene's HBM buffers, a ring of HIP events around the
// render kernels and a ring of device-side segment counters.  There is no CPU rendering path in this
// library: without a device every compute entry point returns RTR_ERR_NO_DEVICE.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h> // types and prototypes only: RCCL is bound at run time (load_rccl)

#include <dlfcn.h>

#include <chrono>
#include <condition_variable>
#include <functional>
#include <memory>
#include <thread>
#include <cstdio>
#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <new>
#include <string>
#include <vector>

#include "../../include/rtr_hip.h"
#include "rtr_device.h"
#include "rtr_scene.h"

namespace rtr {
hipError_t launch_render(const RenderParams &p, uint32_t grid, uint32_t lds_bytes, hipStream_t stream);
hipError_t launch_resolve(const float *sum_rgba, uint32_t spp, uint32_t n_pixels, uint8_t *rgba8, float *mean_rgb,
                          hipStream_t stream);
hipError_t render_kernel_attributes(const RenderParams &p, int *vgprs, int *static_lds);
uint32_t render_resident_waves(const RenderParams &p, uint32_t lds_bytes);
const char *render_kernel_name(const RenderParams &p);
hipError_t launch_sum_samples(const RenderParams &p, const float *samples, uint32_t n_samples, float *acc_io, bool first_batch, bool last_batch, uint32_t spp,
                              uint32_t n_pixels, float *sum_rgba, uint8_t *rgba8, float *mean_rgb, hipStream_t stream);
hipError_t launch_selftest_sqrt(unsigned long long *d_mismatches, hipStream_t stream);
hipError_t launch_debug_fault(unsigned long long *fault, hipStream_t stream);
hipError_t lbvh_build(const float *d_boxes, uint32_t n, const float bounds_min[3], const float bounds_max[3], DevNode *d_nodes_out,
                      uint32_t node_base, uint32_t leaf_base, uint32_t *d_height, hipStream_t stream);
hipError_t launch_wf_generate(const RenderParams &p, uint32_t batch_samples, hipStream_t stream);
hipError_t launch_wf_intersect(const RenderParams &p, uint32_t grid, hipStream_t stream);
hipError_t launch_wf_shade(const RenderParams &p, uint32_t grid, hipStream_t stream);
uint32_t wf_resident_waves(const RenderParams &p, bool shade);
uint32_t wf_intersect_lds_bytes(const RenderParams &p);
uint32_t stack_entry_bytes(const RenderParams &p);
hipError_t wf_intersect_attributes(const RenderParams &p, int *vgprs);
hipError_t launch_selftest_div(float d, uint32_t lo_bits, uint32_t hi_bits, unsigned long long *d_mismatches, hipStream_t stream);
hipError_t launch_sum_partials(const float *const *d_partials, uint32_t n, uint32_t n_pixels, float *d_total, hipStream_t stream);
hipError_t launch_probe_hits(const RenderParams &p, const float *d_origins, const float *d_dirs, uint32_t n, uint32_t *d_out, hipStream_t stream);
} // namespace rtr

#ifndef RTR_BVH_LDS_MATERIALS
#define RTR_BVH_LDS_MATERIALS 16u
#endif

// every entry point that takes a context holds its mutex for the duration of the call
#define RTR_LOCK(ctx) std::lock_guard<std::recursive_mutex> rtr_guard_((ctx)->mu)

namespace {
constexpr int kRing = 64;
thread_local std::string g_create_error;
struct MultiState;
} // namespace

struct rtr_context {
    // rtr_create_multi: the contexts of the GPUs this one spans (empty: an ordinary single-device context) and the
    // communication state; `device` is then the first GPU's, where frames are resolved
    std::vector<rtr_context *> ranks;
    MultiState *multi = nullptr;
    int device = -1; // -1: host-only
    hipStream_t stream = nullptr;
    std::string error;
    rtr::BuildOptions build_options;
    int tree_kernel = 0; // option "tree_kernel": 0 auto, 1 resumable, 2 wave, 3 samples, 4 stream
    uint32_t debug_lds_pad = 0; // option "debug_lds_pad"
    uint32_t block_samples = 8; // option "block_samples": samples of one tile a wave of the samples kernel claims at a time
    uint32_t *d_work = nullptr; // the integrate kernels' ticket counter (one word, zeroed before every launch)
    bool traversal_counters = false; // option "traversal_counters": tree launches take the counting samples kernel
    bool last_counted = false;       // the most recent launch filled d_work[8 ..]: node steps, leaf tests, SkyMap lookups
    bool stack_wide = false;    // option "stack_entries" = "wide"
    uint32_t wf_flush = 8; // option "stream_flush": finished walks a wave of k_wf_intersect collects before storing them
    void *d_wf = nullptr;  // streaming wavefront: the path records of one batch (ray, hit, state, chain, stage)
    size_t wf_cap = 0;
    int sample_chunks = 0; // option "sample_chunks": 0 auto, else the number of pieces
    int chunk_taper = -1;  // option "debug_chunk_taper": -1 auto, else 0..7 (chunk_boundary in rtr_kernels.hip)
    std::vector<uint32_t> chunk_lengths; // option "debug_chunk_lengths": proportions of the chunks ("15:10:6:1"); sets sample_chunks too
    double chunk_timeout_s = 10.0; // option "chunk_timeout_ms": bound of a sample chunk's wait for its predecessor
    bool sky_bilinear = false;     // option "sky_filter"
    bool debug_fault = false;      // option "debug_fault": the next launches set bit 1 of the fault word
    uint64_t debug_alloc_limit = 0; // option "debug_alloc_limit": scratch allocations above this many bytes fail (0: off)
    void *d_samples = nullptr; // k_render_bvh_sp: one float4 per (sample, pixel) of a launch
    uint32_t samples_batch = 0;  // option "samples_batch": samples per batch of that buffer (0: as many as 16 GiB hold)
    size_t samples_cap = 0;
    uint64_t budget_key = 0, budget_bytes = 0; // memory budget of the per-sample buffers, looked up once per launch shape
    void *d_carry = nullptr;       // float4 per pixel: running sums between sample chunks
    size_t carry_cap = 0;
    void *d_chunk_flags = nullptr; // per tile + 1
    size_t chunk_flags_cap = 0;
    mutable std::recursive_mutex mu; // a context is used by one thread at a time; concurrent calls serialise here
    // scene
    bool have_scene = false;
    rtr::FlatScene flat; // host copy (kept for introspection)
    void *d_nodes = nullptr, *d_leaf_hot = nullptr, *d_leaf_cold = nullptr, *d_lin = nullptr, *d_objects = nullptr,
         *d_materials = nullptr, *d_sky = nullptr;
    uint64_t device_bytes = 0;
    double upload_ms = 0.0;
    // per-launch bookkeeping
    hipEvent_t ev_start[kRing] = {}, ev_stop[kRing] = {}, ev_mid[kRing] = {}; // mid: after the launch's last integrate kernel
    std::string kernel_name; // of the most recent launch's integrate kernel
    hipEvent_t ev_res_start = nullptr, ev_res_stop = nullptr;
    bool resolve_timed = false;
    unsigned long long *d_counters = nullptr; // kRing slots + the sticky fault word (slot kRing)
    uint32_t launches = 0;
    uint64_t last_primary = 0;
    uint32_t last_lds = 0, last_vgprs = 0;
    // scratch for the host-output render
    void *d_rgba8 = nullptr; size_t rgba8_cap = 0;
    void *d_mean = nullptr; size_t mean_cap = 0;
};

namespace {

int fail(rtr_context *ctx, int code, const std::string &msg) {
    if (ctx) ctx->error = msg;
    return code;
}
int fail_hip(rtr_context *ctx, hipError_t e, const char *what) {
    return fail(ctx, RTR_ERR_HIP, std::string(what) + ": " + hipGetErrorString(e));
}
#define RTR_HIP_TRY(ctx, call)                                   \
    do {                                                         \
        hipError_t e_ = (call);                                  \
        if (e_ != hipSuccess) return fail_hip(ctx, e_, #call);   \
    } while (0)

void free_scene(rtr_context *ctx) {
    void **bufs[] = {&ctx->d_nodes, &ctx->d_leaf_hot, &ctx->d_leaf_cold, &ctx->d_lin, &ctx->d_objects, &ctx->d_materials, &ctx->d_sky};
    for (void **b : bufs) {
        if (*b) (void)hipFree(*b);
        *b = nullptr;
    }
    ctx->device_bytes = 0;
    ctx->have_scene = false;
}

template <typename T>
int upload_vec(rtr_context *ctx, const std::vector<T> &v, void **dst) {
    *dst = nullptr;
    if (v.empty()) return RTR_OK;
    size_t bytes = v.size() * sizeof(T);
    RTR_HIP_TRY(ctx, hipMalloc(dst, bytes));
    RTR_HIP_TRY(ctx, hipMemcpy(*dst, v.data(), bytes, hipMemcpyHostToDevice));
    ctx->device_bytes += bytes;
    return RTR_OK;
}

// The random streams are keyed by (seed, pixel, sample) (rtr_kernels.hip).  Both halves of the seed go through a hash
// into BOTH keys, so that two seeds never give frames that are pixel permutations of each other (with a plain
// `pixel ^ seed_lo` the streams of (pixel p, seed s) and (pixel p ^ d, seed s ^ d) coincide).
uint32_t hash32(uint32_t x) {
    x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16;
    return x;
}
void seed_keys(uint64_t seed, uint32_t &key0, uint32_t &key1) {
    const uint32_t lo = (uint32_t)seed, hi = (uint32_t)(seed >> 32);
    key0 = hash32(lo ^ hash32(hi + 0x9E3779B9u));
    key1 = hash32(hi ^ hash32(lo + 0x85EBCA6Bu));
}

int check_settings(rtr_context *ctx, const rtr_camera *camera, const rtr_settings *st, uint32_t width, uint32_t height) {
    if (!camera || !st) return fail(ctx, RTR_ERR_INVALID_ARGUMENT, "null camera or settings");
    if (width == 0 || height == 0) return fail(ctx, RTR_ERR_BAD_SETTINGS, "empty render target");
    if ((uint64_t)width * height > 0x7FFFFFFFull) return fail(ctx, RTR_ERR_BAD_SETTINGS, "render target too large");
    if (st->max_ray_depth > 255u) return fail(ctx, RTR_ERR_BAD_SETTINGS, "max_ray_depth is a u8 in the reference (scene.rs:25)");
    if (!(st->clamp_indirect == st->clamp_indirect) || st->clamp_indirect < 0.0f)
        return fail(ctx, RTR_ERR_BAD_SETTINGS, "clamp_indirect NaN or negative: f32::clamp panics (render.rs:95)");
    if (st->mode > RTR_MODE_NORMAL) return fail(ctx, RTR_ERR_BAD_SETTINGS, "unknown render mode");
    return RTR_OK;
}

// The device's sticky fault word (rtr_device.h RenderParams::fault), read where the API has just synchronised.
// A set bit fails the call and is cleared, so one bad launch fails exactly one call.
int check_fault(rtr_context *ctx) {
    unsigned long long word = 0;
    RTR_HIP_TRY(ctx, hipMemcpy(&word, ctx->d_counters + kRing, sizeof word, hipMemcpyDeviceToHost));
    if (!word) return RTR_OK;
    RTR_HIP_TRY(ctx, hipMemset(ctx->d_counters + kRing, 0, sizeof word));
    if (word & 1ull)
        return fail(ctx, RTR_ERR_HIP, "a sample chunk timed out waiting for its predecessor: the frames rendered since the last "
                                      "synchronising call are invalid");
    return fail(ctx, RTR_ERR_HIP, "device fault word set by the debug_fault option");
}

int ensure_scratch(rtr_context *ctx, void **buf, size_t *cap, size_t need) {
    if (*cap >= need) return RTR_OK;
    if (ctx->debug_alloc_limit && need > ctx->debug_alloc_limit) // testing aid: behave as a device that is out of memory
        return fail(ctx, RTR_ERR_OUT_OF_MEMORY, "scratch allocation above debug_alloc_limit");
    if (*buf) (void)hipFree(*buf);
    *buf = nullptr; *cap = 0;
    RTR_HIP_TRY(ctx, hipMalloc(buf, need));
    *cap = need;
    return RTR_OK;
}

// The scene's part of the launch parameters: which world render.rs:68-72 selects and where its records live in HBM.
void set_world_params(const rtr_context *ctx, bool enable_bvh_tree, rtr::RenderParams &p) {
    const rtr::FlatScene &fs = ctx->flat;
    // render.rs:68-72: the tree when enable_bvh_tree, else the linear scan of scene.rs:141-157
    // stack entries: a walk pushes at most once per node record on its path (FlatScene::stack_depth, checked by
    // tests/cpp/scene_sanity.cpp); one more where a Volume's nested mesh walk starts above the outer walk's entries.
    // Exact, because LDS per wave bounds the tree kernels' occupancy: house.obj 6016 -> 5632 B = 27 -> 28 waves per CU.
    if (enable_bvh_tree) { p.world = rtr::WORLD_BVH; p.stack_cap = std::max(1u, fs.stack_depth + (fs.has_volumes ? 1u : 0u)); }
    else if (fs.spheres_only) { p.world = rtr::WORLD_LINEAR_SPHERES; p.stack_cap = 0u; }
    else { p.world = rtr::WORLD_LINEAR_GENERIC; p.stack_cap = fs.mesh_stack_depth + 1u; }
    p.n_objects = fs.n_objects;
    p.has_volumes = fs.has_volumes ? ((fs.has_mesh_volumes || fs.has_nested_volumes) ? 2u : 1u) : 0u; // 2: the kernels with the nested mesh walk and nested Volumes
    p.has_meshes = fs.n_triangles > 0u ? 1u : 0u;
    p.root_ref = fs.root_ref;
    std::memcpy(p.root_box, fs.root_box, sizeof p.root_box);
    p.nodes = static_cast<const rtr::DevNode *>(ctx->d_nodes);
    p.leaf_hot = static_cast<const rtr::DevLeafHot *>(ctx->d_leaf_hot);
    p.leaf_cold = static_cast<const rtr::DevLeafCold *>(ctx->d_leaf_cold);
    p.lin_spheres = static_cast<const rtr::DevLinSphere *>(ctx->d_lin);
    p.objects = static_cast<const rtr::DevObject *>(ctx->d_objects);
    p.materials = static_cast<const rtr::DevMaterial *>(ctx->d_materials);
    p.n_materials = (uint32_t)fs.materials.size();
    p.bg_kind = fs.bg_kind;
    std::memcpy(p.bg_a, fs.bg_a, sizeof p.bg_a);
    std::memcpy(p.bg_b, fs.bg_b, sizeof p.bg_b);
    p.sky = static_cast<const float *>(ctx->d_sky);
    p.sky_w = fs.sky_w; p.sky_h = fs.sky_h;
    p.sky_bilinear = ctx->sky_bilinear ? 1u : 0u;
}

// Fill the launch parameters and launch the integrator.  Exactly one of (sum_rgba) / (rgba8|mean_rgb) is used.
int launch_frame(rtr_context *ctx, const rtr_camera *camera, const rtr_settings *st, uint32_t width, uint32_t height,
                 uint64_t seed, uint32_t shard, uint32_t n_shards, float *d_sum_rgba, uint8_t *d_rgba8, float *d_mean_rgb,
                 hipStream_t stream, uint32_t sample_begin = 0u, uint32_t sample_count = 0xFFFFFFFFu) {
    if (!ctx) return RTR_ERR_INVALID_ARGUMENT;
    if (ctx->device < 0) return fail(ctx, RTR_ERR_NO_DEVICE, "host-only context: no HIP device bound (there is no CPU fallback)");
    if (!ctx->have_scene) return fail(ctx, RTR_ERR_NO_SCENE, "rtr_upload_scene has not been called");
    int rc = check_settings(ctx, camera, st, width, height);
    if (rc) return rc;
    if (n_shards == 0 || shard >= n_shards) return fail(ctx, RTR_ERR_INVALID_ARGUMENT, "shard must be < n_shards");
    RTR_HIP_TRY(ctx, hipSetDevice(ctx->device));
    if (!stream) stream = ctx->stream;

    const rtr::FlatScene &fs = ctx->flat;
    rtr::RenderParams p;
    std::memset(&p, 0, sizeof p);
    rtr::camera_ray_origin(*camera, p.cam_origin, p.cam_horizontal, p.cam_vertical, p.cam_llc);
    p.width_m1 = (float)(width - 1u);   // (target.width - 1) as Float, render.rs:88
    p.height_m1 = (float)(height - 1u); // render.rs:89
    p.width = width; p.height = height;
    p.spp = st->samples_per_pixel; p.max_depth = st->max_ray_depth; p.mode = st->mode;
    if (sample_count == 0xFFFFFFFFu) { sample_begin = 0u; sample_count = p.spp; }
    if (sample_begin > p.spp || sample_count > p.spp - sample_begin)
        return fail(ctx, RTR_ERR_INVALID_ARGUMENT, "sample range exceeds samples_per_pixel");
    p.sample_begin = sample_begin; p.sample_count = sample_count;
    p.clamp_indirect = st->clamp_indirect;
    seed_keys(seed, p.seed_key0, p.seed_key1);
    set_world_params(ctx, st->enable_bvh_tree != 0u, p);
    // "auto": the resumable walk (k_render_bvh) for scenes with meshes.  Measured against the wave-synchronous
    // k_render<BVH>: +45 % on house.obj, +10 % on monkey.obj; on sphere-only trees equal within 2 % (reference trees)
    // or 7 % behind (SAH trees), so those keep k_render.
    p.resumable_walk = (ctx->tree_kernel == 1 || (ctx->tree_kernel == 0 && fs.n_triangles > 0u)) ? 1u : 0u;
    // "samples" (and "auto" whenever it applies): lanes take (sample, pixel) pairs and each sample's radiance goes
    // through HBM (k_render_bvh_sp + k_sum_samples).  Needs a tree world, something to trace and the fused or
    // sample-sharded path (a tile shard writes into a frame other shards share); otherwise the resumable / wave rule.
    // Measured at 1080p/64 spp against the resumable walk: C3 +19 %, C4 +16 %, C1 +10 %.
    // Tiny trees keep the wave kernel under "auto": the per-sample set-up and the trip through memory do not pay for
    // their cheap samples (5-sphere tree -13 %; 102 spheres +8 % at 1080p, +1 % at 4K; meshes +20 %).
    const bool per_sample_ok = st->enable_bvh_tree && p.max_depth > 0u && sample_count > 0u && n_shards == 1u;
    // "stream": path state in HBM, one kernel per stage and depth (rtr_kernels.hip "streaming wavefront"); same conditions
    const bool streaming = ctx->tree_kernel == 4 && per_sample_ok && !ctx->traversal_counters;
    p.sample_parallel = ((ctx->tree_kernel == 3 || (ctx->tree_kernel == 0 && fs.leaf_hot.size() >= 32u) || ctx->traversal_counters) && per_sample_ok) ? 1u : 0u;
    if ((ctx->tree_kernel == 3 || ctx->tree_kernel == 4) && !p.sample_parallel && !streaming) p.resumable_walk = 1u;
    if (streaming) p.sample_parallel = 1u; // shares the per-sample radiance buffer and k_sum_samples with "samples"
    p.tiles_x = (width + 7u) / 8u;
    uint32_t tiles_y = (height + 7u) / 8u;
    p.tiles_total = p.tiles_x * tiles_y;
    p.shard = shard; p.n_shards = n_shards;
    p.sum_rgb = d_sum_rgba; p.rgba8 = d_rgba8; p.mean_rgb = d_mean_rgb;
    const uint32_t slot = ctx->launches % kRing;
    p.counter = ctx->d_counters + slot;
    p.fault = ctx->d_counters + kRing;
    p.chunk_timeout_ticks = (unsigned long long)(ctx->chunk_timeout_s * 1.0e8); // s_memrealtime counts at 100 MHz

    uint32_t grid = p.tiles_total > shard ? (p.tiles_total - shard + n_shards - 1u) / n_shards : 0u;
    // small tables staged in LDS: the linear sphere world's records, and the materials when they fit 4 KB
    p.sphere_regs = (p.world == rtr::WORLD_LINEAR_SPHERES && fs.n_objects >= 1u && fs.n_objects <= 8u) ? fs.n_objects : 0u;
    // materials go to LDS when that does not cost occupancy: always for the register-bound sphere kernel, only
    // small tables for the latency-bound tree kernels (their traversal stack already takes ~8 KB per wave)
    const uint32_t lds_mat_cap = (p.world == rtr::WORLD_LINEAR_SPHERES) ? 128u : RTR_BVH_LDS_MATERIALS;
    p.lds_materials = (p.n_materials > 0u && p.n_materials <= lds_mat_cap) ? p.n_materials : 0u;
    // dynamic LDS per wave: attenuation chain + traversal stack (refs + entry distances) + the tables
    // The attenuation chain: a column of max_ray_depth entries per lane in LDS, or -- in the tree worlds, when
    // max_ray_depth ids fit 32 bits (the all-ones code is the Clay id) -- one register.  The tree kernels' occupancy is
    // bounded by LDS, so giving the chain's region back pays (house.obj: 6.8 -> 6.0 KB per wave, C4 +5 %); the sphere
    // world has LDS to spare and the register form's variable shifts cost it 3 % (measured), so it keeps the LDS column.
    {
        uint32_t width = 1u;
        while (width < 32u && ((1u << width) - 1u) < p.n_materials) width++;
        p.chain_bits = (p.world != rtr::WORLD_LINEAR_SPHERES && width * p.max_depth <= 32u && width < 32u) ? width : 0u;
        p.chain_lds = p.chain_bits ? 0u : p.max_depth;
        if (p.world == rtr::WORLD_LINEAR_SPHERES) { // the attenuations themselves (three words per entry) while that stays small
            if (p.max_depth <= 16u) p.chain_lds = 3u * p.max_depth;
            else p.chain_bits = 0xFFFFFFFFu; // ids
        }
    }
    // one-word stack entries when every index fits 16 bits (rtr_kernels.hip stack_pack32); option "stack_entries" = "wide" keeps
    // the general form for tests
    {   // the sphere world's plain case (benchmark.json): k_render's instantiation without the mode / background / mix tests
        bool plain = p.sphere_regs >= 1u && p.mode == 0u && p.bg_kind == 0u && p.chain_bits != 0xFFFFFFFFu && p.lds_materials > 0u;
        for (const rtr::DevMaterial &m : fs.materials) plain = plain && m.kind != 4u && m.kind != 5u;
        p.plain_sphere_world = plain ? 1u : 0u;
    }
    p.stack_packed = (p.sample_parallel && !streaming && !ctx->stack_wide && fs.nodes.size() <= 65536u && fs.leaf_hot.size() <= 65536u) ? 1u : 0u; // (the samples kernel has both forms)
    uint32_t lds_bytes = p.chain_lds * 64u * 4u + p.stack_cap * 64u * rtr::stack_entry_bytes(p) + (2u * p.sphere_regs + p.lds_materials) * 32u; // spheres, their materials' copies, the material table
    lds_bytes += ctx->debug_lds_pad;
    if (lds_bytes > 160u * 1024u) return fail(ctx, RTR_ERR_UNSUPPORTED, "max_ray_depth + tree height exceed one CU's LDS");
    ctx->last_lds = lds_bytes;
    ctx->last_primary = 0;
    // this shard's share of W*H*spp
    {
        uint64_t pixels = 0;
        for (uint32_t t = shard; t < p.tiles_total; t += n_shards) {
            uint32_t tx = t % p.tiles_x, ty = t / p.tiles_x;
            uint32_t w = (tx * 8u + 8u <= width) ? 8u : width - tx * 8u;
            uint32_t h = (ty * 8u + 8u <= height) ? 8u : height - ty * 8u;
            pixels += (uint64_t)w * h;
        }
        ctx->last_primary = pixels * sample_count;
    }
    // Sample chunks (rtr_kernels.hip "Tickets and sample chunks").  They only pay when the launch is short of work items --
    // at least two tiles per resident wave, but fewer than ~24 -- so 4K frames take one piece.  Sphere world (round 3, with
    // the hand-over made of device-scope atomics): few pieces of steeply falling length -- a workgroup's ragged end costs
    // about as much for a long piece as for a short one, and only the LAST pieces are what the chip drains on.  Headline
    // frame, proportions a:b:c of the samples: 24:7:1 13.01 ms, 28:3:1 13.00, 20:10:2 13.06, 15:10:6:1 (linear taper 7) 13.11,
    // 16:8:4:2:1 13.32, 15:1 13.74, one piece 14.72; 128 samples 24:7:1 6.65 / 7:1 6.73 / one piece 7.41; 64 samples 7:1 3.49 / 24:7:1 3.66 / one
    // piece 3.80; 32 samples 7:1 1.87 / one piece 2.02; 24 samples 1.69 / 1.68; 16 samples 1.34 / 1.17 (what a rank of the 2-, 4- and 8-GPU sample split renders).  Tree worlds
    // (k_render_bvh): up to 4 equal pieces of at least 16 samples.
    p.tiles_launch = grid > 0u ? grid : 1u;
    p.chunks = 1u;
    if (grid > 0u) {
        uint32_t want = 1u;
        int auto_profile = 0; // sphere world, automatic choice: 1 = proportions 24:7:1, 2 = 7:1
        if (ctx->sample_chunks > 0) {
            want = (uint32_t)ctx->sample_chunks;
        } else if (p.sample_parallel) { // independent work items, no minimum length to speak of: ~80 per resident wave
            // (1080p/256 spp, C3 / C4: 1 chunk 152 / 135 ms, 2: 128 / 115, 4: 119 / 106, 8: 115 / 102, 16: 115 / 102, 32: 119 / 107)
            const uint32_t resident = std::max(1u, rtr::render_resident_waves(p, lds_bytes));
            want = std::min(std::max(1u, 80u * resident / grid), std::max(1u, sample_count / 4u));
        } else {
            const uint32_t resident = std::max(1u, rtr::render_resident_waves(p, lds_bytes));
            const uint32_t by_length = p.world == rtr::WORLD_LINEAR_SPHERES ? (sample_count >= 96u ? 3u : sample_count >= 28u ? 2u : 1u)
                                                                            : std::min(4u, sample_count / 16u);
            if (grid >= 2u * resident) want = std::min(24u * resident / grid, by_length);
            if (p.world == rtr::WORLD_LINEAR_SPHERES) auto_profile = want == 3u ? 1 : want == 2u ? 2 : 0;
        }
        want = std::max(1u, std::min(want, std::max(1u, sample_count)));
        if ((uint64_t)grid * want > 0x7FFFFFFFull) want = 1u;
        p.chunks = want;
        // an explicit chunk count ("sample_chunks"): linear taper, chunk_boundary (headline frame, 4 chunks: taper 4 / 5 / 6 / 7
        // 13.47 / 13.32 / 13.21 / 13.08 ms)
        p.chunk_taper = p.world == rtr::WORLD_LINEAR_SPHERES ? 7u : 0u;
        if (ctx->chunk_taper >= 0) p.chunk_taper = (uint32_t)ctx->chunk_taper;
        static const std::vector<uint32_t> kThree = {24u, 7u, 1u}, kTwo = {7u, 1u};
        const std::vector<uint32_t> &lengths = ctx->chunk_lengths.size() == want ? ctx->chunk_lengths
                                               : (auto_profile == 1 && want == 3u) ? kThree
                                               : (auto_profile == 2 && want == 2u) ? kTwo : ctx->chunk_lengths;
        // the boundaries of up to eight chunks go to the kernel as a table (chunk_boundary in rtr_kernels.hip: the same
        // linear taper; any boundaries give the same frame), or come from option "debug_chunk_lengths"
        p.chunk_table_on = 0u;
        if (want <= 8u && !p.sample_parallel) {
            const bool given = lengths.size() == want;
            uint64_t sum_given = 0;
            for (uint32_t v : lengths) sum_given += v;
            for (uint32_t c = 0; c <= want; c++) {
                uint64_t first;
                if (given) {
                    uint64_t before = 0;
                    for (uint32_t i = 0; i < c; i++) before += lengths[i];
                    first = (uint64_t)sample_count * before / std::max<uint64_t>(sum_given, 1ull); // lengths are proportions
                } else if (p.chunk_taper == 0u || want < 2u) {
                    first = (uint64_t)sample_count * c / want;
                } else {
                    const uint64_t k = p.chunk_taper, n = want;
                    const uint64_t num = (uint64_t)c * (8u + k) * (n - 1u) - k * (uint64_t)c * (c - 1u);
                    first = (uint64_t)sample_count * num / (8ull * n * (n - 1u));
                }
                p.chunk_first[c] = (uint32_t)first;
            }
            p.chunk_first[want] = sample_count;
            p.chunk_table_on = 1u;
        }
    }
    // Per-sample radiance buffer of the "samples" and "stream" tree kernels: sample_count * W * H float4, filled in
    // batches ("stream" adds its path records: 65 + 4 * max_ray_depth bytes per path of the batch).  A batch takes at
    // most half of the device memory that is free right now and at most 16 GiB of radiance; any batch size gives the
    // same frame (the running sums are carried from batch to batch), so an allocation that fails is retried with half
    // the batch before the call gives up.
    uint32_t batch_samples = sample_count;
    uint64_t wf_paths_per_sample = 0, wf_bytes_per_path = 0;
    if (p.sample_parallel) {
        const uint64_t per_sample = (uint64_t)width * height * 4ull * sizeof(float);
        uint64_t per_sample_total = per_sample;
        if (streaming) {
            wf_paths_per_sample = (uint64_t)p.tiles_total * 64ull;
            wf_bytes_per_path = 32ull + 16ull + 16ull + 4ull * p.max_depth + 1ull;
            per_sample_total += wf_paths_per_sample * wf_bytes_per_path;
        }
        // The memory budget is looked up when the launch shape changes (or after an allocation failure), not per frame:
        // hipMemGetInfo inside every launch made the batch depend on what else the process had allocated meanwhile (torch's
        // caching allocator, say), and a batch that grew mid-run frees and reallocates the buffer -- a device synchronisation
        // in the middle of a pipelined run.  In steady state the batch never changes.
        const uint64_t shape_key = ((uint64_t)width << 40) ^ ((uint64_t)height << 20) ^ ((uint64_t)sample_count << 8) ^ ((uint64_t)p.max_depth << 1) ^ (streaming ? 1ull : 0ull);
        if (ctx->budget_key != shape_key || ctx->budget_bytes == 0) {
            size_t free_b = 0, total_b = 0;
            uint64_t fresh = 16ull << 30;
            if (hipMemGetInfo(&free_b, &total_b) == hipSuccess) {
                const uint64_t held = (uint64_t)ctx->samples_cap + (uint64_t)ctx->wf_cap; // ours already: reusable
                fresh = ((uint64_t)free_b + held) / 2ull;
            }
            ctx->budget_key = shape_key;
            ctx->budget_bytes = std::max<uint64_t>(fresh, 1ull);
        }
        const uint64_t budget = std::max<uint64_t>(per_sample_total, ctx->budget_bytes);
        uint64_t fit = std::max<uint64_t>(1ull, std::min<uint64_t>(budget / per_sample_total, (16ull << 30) / per_sample));
        fit = std::max<uint64_t>(1ull, std::min<uint64_t>(fit, 0xFFFFFFFFull / ((uint64_t)width * height))); // 32-bit (sample, pixel) slots in the kernels
        if (streaming) fit = std::min<uint64_t>(fit, std::min<uint64_t>(0x7FFFFFFFull / wf_paths_per_sample, 65535ull));
        if (streaming && fit > 64ull) fit = 64ull; // more samples per batch buy nothing: a batch is ~15 launches
        batch_samples = (uint32_t)std::max<uint64_t>(1ull, std::min<uint64_t>(sample_count, fit));
        if (ctx->samples_batch > 0u) batch_samples = std::min(batch_samples, ctx->samples_batch);
        for (;;) {
            int rc = ensure_scratch(ctx, &ctx->d_samples, &ctx->samples_cap, (size_t)(per_sample * batch_samples));
            if (!rc && streaming) rc = ensure_scratch(ctx, &ctx->d_wf, &ctx->wf_cap, (size_t)(wf_paths_per_sample * batch_samples * wf_bytes_per_path + 256ull));
            if (!rc) break;
            if (batch_samples == 1u) return rc;
            (void)hipGetLastError(); // out of memory: the same frame in smaller batches
            batch_samples = std::max(1u, batch_samples / 2u);
            ctx->budget_bytes = std::max<uint64_t>(1ull, per_sample_total * batch_samples); // remembered for this shape
        }
        p.samples = static_cast<float *>(ctx->d_samples);
        if (batch_samples < sample_count) { // running sums between batches
            int rc = ensure_scratch(ctx, &ctx->d_carry, &ctx->carry_cap, (size_t)per_sample);
            if (rc) return rc;
        }
    }
    if (p.chunks > 1u && !p.sample_parallel) {
        int rc = ensure_scratch(ctx, &ctx->d_carry, &ctx->carry_cap, (size_t)width * height * 4u * sizeof(float));
        if (rc) return rc;
        if ((rc = ensure_scratch(ctx, &ctx->d_chunk_flags, &ctx->chunk_flags_cap, (size_t)grid * sizeof(uint32_t)))) return rc;
        p.carry_rgba = static_cast<float *>(ctx->d_carry);
        p.chunk_flags = static_cast<uint32_t *>(ctx->d_chunk_flags);
        RTR_HIP_TRY(ctx, hipMemsetAsync(ctx->d_chunk_flags, 0, (size_t)grid * sizeof(uint32_t), stream));
    }
    RTR_HIP_TRY(ctx, hipMemsetAsync(ctx->d_counters + slot, 0, sizeof(unsigned long long), stream));
    ctx->last_counted = ctx->traversal_counters && p.sample_parallel && !streaming && grid > 0;
    if (ctx->last_counted) { // the counting instantiation of the samples kernel adds into three words behind the ticket counter
        p.walk_counters = reinterpret_cast<unsigned long long *>(ctx->d_work) + 1;
        RTR_HIP_TRY(ctx, hipMemsetAsync(p.walk_counters, 0, 3 * sizeof(unsigned long long), stream));
    }
    RTR_HIP_TRY(ctx, hipEventRecord(ctx->ev_start[slot], stream));
    if (grid > 0 && !p.sample_parallel) {
        // One workgroup (one wave) per ticket -- (chunk, tile), chunk-major; a workgroup claims its ticket from the counter
        // when it starts running (rtr_kernels.hip "Tickets and sample chunks": forward progress without any dispatch order).
        p.work_counter = ctx->d_work;
        p.work_blocks = p.tiles_launch * p.chunks;
        RTR_HIP_TRY(ctx, hipMemsetAsync(ctx->d_work, 0, sizeof(uint32_t), stream));
        RTR_HIP_TRY(ctx, rtr::launch_render(p, p.work_blocks, lds_bytes, stream));
    }
    if (!p.sample_parallel) RTR_HIP_TRY(ctx, hipEventRecord(ctx->ev_mid[slot], stream));
    if (grid == 0 && p.sample_parallel) RTR_HIP_TRY(ctx, hipEventRecord(ctx->ev_mid[slot], stream));
    if (grid > 0 && p.sample_parallel) {
        // batch by batch: render the batch's samples, then add them to each pixel's sum in index order; the last
        // batch's k_sum_samples finishes the pixel (or stores the launch's partial sum)
        uint32_t wf_grid_i = 0, wf_grid_s = 0;
        if (streaming) {
            p.wf_flush = ctx->wf_flush;
            wf_grid_i = std::max(1u, rtr::wf_resident_waves(p, false));
            wf_grid_s = std::max(1u, rtr::wf_resident_waves(p, true));
            ctx->last_lds = rtr::wf_intersect_lds_bytes(p);
        }
        for (uint32_t done = 0; done < sample_count; done += batch_samples) {
            rtr::RenderParams b = p;
            b.sample_begin = sample_begin + done;
            b.sample_count = std::min(batch_samples, sample_count - done);
            b.chunks = std::max(1u, std::min(p.chunks, b.sample_count));
            const bool first = done == 0u, last = done + b.sample_count == sample_count;
            if (!streaming) {
                // one workgroup per wave the device keeps resident; the waves claim blocks of (tile, block_samples samples) from a
                // counter (measured at 1080p / 256 spp: 1: +45 %, 2: +3 ... 8 %, 4: +1 %, 8: best, 16: +1 %, 32: +5 %, 64: +13 %)
                RTR_HIP_TRY(ctx, hipMemsetAsync(ctx->d_work, 0, sizeof(uint32_t), stream));
                b.work_counter = ctx->d_work;
                b.samples_per_block = ctx->block_samples;
                const uint64_t blocks = (uint64_t)grid * ((b.sample_count + b.samples_per_block - 1u) / b.samples_per_block);
                if (blocks > 0xFFFFFFF0ull) return fail(ctx, RTR_ERR_UNSUPPORTED, "too many work blocks in one batch: lower samples_batch");
                b.work_blocks = (uint32_t)blocks;
                b.chunks = 1u;
                const uint32_t resident = std::max(1u, rtr::render_resident_waves(b, lds_bytes));
                RTR_HIP_TRY(ctx, rtr::launch_render(b, (uint32_t)std::min<uint64_t>(resident, blocks), lds_bytes, stream));
            } else {
                // the batch's path records, carved out of one allocation (16-byte aligned pieces first)
                const uint64_t n = wf_paths_per_sample * b.sample_count;
                uint8_t *base = static_cast<uint8_t *>(ctx->d_wf);
                b.wf_ray = reinterpret_cast<float *>(base); base += n * 32ull;
                b.wf_hit = reinterpret_cast<float *>(base); base += n * 16ull;
                b.wf_state = reinterpret_cast<uint32_t *>(base); base += n * 16ull;
                b.wf_chain = reinterpret_cast<uint32_t *>(base); base += n * 4ull * p.max_depth;
                b.wf_stage = base;
                b.wf_paths = (uint32_t)n;
                b.wf_blocks = (uint32_t)((n + 63ull) / 64ull);
                RTR_HIP_TRY(ctx, rtr::launch_wf_generate(b, b.sample_count, stream));
                for (uint32_t d = 0; d < p.max_depth; d++) { // trace_ray's recursion levels (render.rs:32-36)
                    b.wf_depth = d;
                    RTR_HIP_TRY(ctx, rtr::launch_wf_intersect(b, std::min(wf_grid_i, b.wf_blocks), stream));
                    RTR_HIP_TRY(ctx, rtr::launch_wf_shade(b, std::min(wf_grid_s, b.wf_blocks), stream));
                }
                p.wf_ray = b.wf_ray; // (for render_kernel_name below)
            }
            if (last) RTR_HIP_TRY(ctx, hipEventRecord(ctx->ev_mid[slot], stream));
            RTR_HIP_TRY(ctx, rtr::launch_sum_samples(p, p.samples, b.sample_count, static_cast<float *>(ctx->d_carry), first, last, p.spp,
                                                     width * height, d_sum_rgba, d_rgba8, d_mean_rgb, stream));
        }
    }
    if (ctx->debug_fault) RTR_HIP_TRY(ctx, rtr::launch_debug_fault(p.fault, stream));
    RTR_HIP_TRY(ctx, hipEventRecord(ctx->ev_stop[slot], stream));
    ctx->launches++;
    {
        int vg = 0, sl = 0;
        if (p.wf_ray ? rtr::wf_intersect_attributes(p, &vg) == hipSuccess : rtr::render_kernel_attributes(p, &vg, &sl) == hipSuccess)
            ctx->last_vgprs = (uint32_t)vg;
    }
    ctx->kernel_name = rtr::render_kernel_name(p);
    return RTR_OK;
}


// The device half of rtr_upload_scene: copy a flattened scene to the context's GPU (and run the device-side tree builder
// where the scene asks for one); `flat` is consumed.
int install_scene(rtr_context *ctx, rtr::FlatScene &flat, std::chrono::steady_clock::time_point t0) {
    int rc = RTR_OK;
    if (ctx->device < 0 && !flat.lbvh_jobs.empty())
        return fail(ctx, RTR_ERR_NO_DEVICE, "mesh_tree = \"lbvh\" builds the mesh trees on the device: not available in a host-only context");
    if (ctx->device >= 0) {
        RTR_HIP_TRY(ctx, hipSetDevice(ctx->device));
        RTR_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
        free_scene(ctx);
        if ((rc = upload_vec(ctx, flat.nodes, &ctx->d_nodes))) return rc;
        if (!flat.lbvh_jobs.empty()) { // device-side builder (rtr_lbvh.hip): writes the reserved node records in place
            void *d_boxes = nullptr;
            uint32_t *d_height = nullptr;
            uint64_t scratch = ctx->device_bytes;
            if ((rc = upload_vec(ctx, flat.lbvh_boxes, &d_boxes))) return rc;
            ctx->device_bytes = scratch; // (freed below: not part of the resident scene)
            hipError_t e = hipMalloc((void **)&d_height, sizeof(uint32_t));
            if (e == hipSuccess) e = hipMemset(d_height, 0, sizeof(uint32_t));
            for (size_t j = 0; e == hipSuccess && j < flat.lbvh_jobs.size(); j++) {
                const rtr::LbvhJob &job = flat.lbvh_jobs[j];
                e = rtr::lbvh_build(static_cast<const float *>(d_boxes) + 6ull * job.box_offset, job.n_tris, job.bounds_min, job.bounds_max,
                                    static_cast<rtr::DevNode *>(ctx->d_nodes) + job.node_base, job.node_base, job.leaf_base, d_height, ctx->stream);
            }
            uint32_t height = 0;
            if (e == hipSuccess) e = hipMemcpy(&height, d_height, sizeof height, hipMemcpyDeviceToHost);
            if (e == hipSuccess) e = hipMemcpy(flat.nodes.data(), ctx->d_nodes, flat.nodes.size() * sizeof(rtr::DevNode), hipMemcpyDeviceToHost);
            if (d_boxes) (void)hipFree(d_boxes);
            if (d_height) (void)hipFree(d_height);
            if (e != hipSuccess) return fail_hip(ctx, e, "device-side BVH build");
            // the walk's stack bound: the host part counted a device-built subtree as a leaf
            flat.stack_depth += height;
            flat.tree_height += height;
            flat.mesh_stack_depth = std::max(flat.mesh_stack_depth, height);
            flat.mesh_height = std::max(flat.mesh_height, height);
            flat.lbvh_boxes.clear();
            flat.lbvh_boxes.shrink_to_fit();
        }
        if ((rc = upload_vec(ctx, flat.leaf_hot, &ctx->d_leaf_hot))) return rc;
        if ((rc = upload_vec(ctx, flat.leaf_cold, &ctx->d_leaf_cold))) return rc;
        if ((rc = upload_vec(ctx, flat.lin_spheres, &ctx->d_lin))) return rc;
        if ((rc = upload_vec(ctx, flat.objects, &ctx->d_objects))) return rc;
        if ((rc = upload_vec(ctx, flat.materials, &ctx->d_materials))) return rc;
        if ((rc = upload_vec(ctx, flat.sky, &ctx->d_sky))) return rc;
    }
    flat.sky.clear(); // the texels live in HBM only
    flat.sky.shrink_to_fit();
    ctx->flat = std::move(flat);
    ctx->have_scene = true;
    ctx->upload_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    return RTR_OK;
}


// ---------------------------------------------------------------------------------------------------------------------
// One context over several GPUs (rtr_create_multi): render() -- lib/core/render.rs:61,114-119,
// the reference's rayon fan-out over pixels -- becomes a fan-out over devices.  The scene is replicated; a frame is split
// by SAMPLES (every rank integrates all pixels for a contiguous share of the samples: the random streams are keyed by the
// absolute sample index, so the ranks draw exactly the single-GPU numbers and every GPU keeps a full grid of waves) or by
// TILES (option "multi_split"); the ranks' partial sums (three packed floats per pixel) are summed onto rank 0 with ONE ncclReduce over xGMI per
// frame and rank 0 runs k_resolve (sum / spp, sqrt gamma, RGBA8: render.rs:103-110 must follow the reduce).  One host
// thread per GPU issues that GPU's work (its own compute and communication streams); partial-sum buffers are double
// buffered, so frame i's reduce + resolve overlap frame i+1's integrate.
// RCCL is bound at run time (dlopen of librccl.so.1: the copy the process already holds -- PyTorch's -- or ROCm's), so the
// library carries no link-time dependency on it and single-GPU contexts never load it.
struct RcclApi {
    void *handle = nullptr;
    decltype(&ncclCommInitAll) CommInitAll = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclReduce) Reduce = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
};
bool load_rccl(RcclApi &api, std::string &err) {
    static std::mutex mu;
    static RcclApi cached;
    std::lock_guard<std::mutex> g(mu);
    if (!cached.handle) {
        for (const char *name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
            cached.handle = dlopen(name, RTLD_NOW | RTLD_LOCAL);
            if (cached.handle) break;
        }
        if (!cached.handle) { err = std::string("cannot load RCCL (librccl.so.1): ") + dlerror(); return false; }
        cached.CommInitAll = reinterpret_cast<decltype(&ncclCommInitAll)>(dlsym(cached.handle, "ncclCommInitAll"));
        cached.CommDestroy = reinterpret_cast<decltype(&ncclCommDestroy)>(dlsym(cached.handle, "ncclCommDestroy"));
        cached.Reduce = reinterpret_cast<decltype(&ncclReduce)>(dlsym(cached.handle, "ncclReduce"));
        cached.GetErrorString = reinterpret_cast<decltype(&ncclGetErrorString)>(dlsym(cached.handle, "ncclGetErrorString"));
        if (!cached.CommInitAll || !cached.CommDestroy || !cached.Reduce || !cached.GetErrorString) {
            err = "librccl.so.1 lacks ncclCommInitAll / ncclReduce";
            cached = RcclApi();
            return false;
        }
    }
    api = cached;
    return true;
}

// One host thread per GPU: runs the jobs it is handed, in order.
class RankWorker {
  public:
    RankWorker() : th_([this] { loop(); }) {}
    ~RankWorker() {
        { std::lock_guard<std::mutex> g(mu_); stop_ = true; }
        cv_.notify_all();
        th_.join();
    }
    void post(std::function<void()> job) {
        { std::lock_guard<std::mutex> g(mu_); job_ = std::move(job); busy_ = true; }
        cv_.notify_all();
    }
    void wait() {
        std::unique_lock<std::mutex> g(mu_);
        cv_.wait(g, [this] { return !busy_; });
    }

  private:
    void loop() {
        std::unique_lock<std::mutex> g(mu_);
        for (;;) {
            cv_.wait(g, [this] { return stop_ || busy_; });
            if (stop_) return;
            std::function<void()> job = std::move(job_);
            g.unlock();
            job();
            g.lock();
            busy_ = false;
            cv_.notify_all();
        }
    }
    std::mutex mu_;
    std::condition_variable cv_;
    std::function<void()> job_;
    bool busy_ = false, stop_ = false;
    std::thread th_;
};

struct MultiState {
    bool same_device = false;      // every rank on one GPU (test configuration): the partial sums are added by a kernel, no RCCL
    bool split_tiles = false;      // option "multi_split" = "tiles"
    RcclApi rccl;
    std::vector<ncclComm_t> comms; // one per rank (distinct devices only)
    std::vector<std::unique_ptr<RankWorker>> workers;
    std::vector<hipStream_t> comm_stream;              // per rank, beside the rank context's compute stream
    std::vector<hipEvent_t> ev_integrated[2], ev_reduced[2]; // per buffer, per rank
    std::vector<void *> d_sum[2];  // per buffer, per rank: partial sums of a frame, three packed floats per pixel
    void *d_total[2] = {nullptr, nullptr}; // rank 0: the reduced sums
    size_t sum_cap = 0;
    void *d_ptrs = nullptr;        // same_device: the ranks' buffer addresses for the adding kernel
    uint64_t frames = 0;
    std::vector<int> rank_rc;      // status of each rank's last job
};

// rank r's samples of a frame: contiguous, the first spp % n ranks take one more (raytracingrust_amd/distributed.py sample_range)
void multi_sample_range(uint32_t rank, uint32_t n, uint32_t spp, uint32_t &begin, uint32_t &count) {
    const uint32_t base = spp / n, extra = spp % n;
    begin = rank * base + std::min(rank, extra);
    count = base + (rank < extra ? 1u : 0u);
}

int multi_fail(rtr_context *ctx, const rtr_context *rank, int rc, uint32_t r) {
    return fail(ctx, rc, "GPU " + std::to_string(rank->device) + " (rank " + std::to_string(r) + "): " + rank->error);
}

int multi_upload(rtr_context *ctx, rtr::FlatScene &flat, std::chrono::steady_clock::time_point t0) {
    const size_t n = ctx->ranks.size();
    std::vector<rtr::FlatScene> copies(n);
    for (size_t r = 0; r + 1 < n; r++) copies[r] = flat;
    copies[n - 1] = std::move(flat);
    MultiState &m = *ctx->multi;
    for (size_t r = 0; r < n; r++) {
        rtr_context *rk = ctx->ranks[r];
        rk->sky_bilinear = ctx->sky_bilinear;
        m.workers[r]->post([&m, rk, &copies, r, t0] { std::lock_guard<std::recursive_mutex> g(rk->mu); m.rank_rc[r] = install_scene(rk, copies[r], t0); });
    }
    for (size_t r = 0; r < n; r++) m.workers[r]->wait();
    for (size_t r = 0; r < n; r++)
        if (m.rank_rc[r]) return multi_fail(ctx, ctx->ranks[r], m.rank_rc[r], (uint32_t)r);
    ctx->have_scene = true;
    ctx->upload_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    return RTR_OK;
}

void multi_free_buffers(rtr_context *ctx) {
    MultiState &m = *ctx->multi;
    for (size_t r = 0; r < ctx->ranks.size(); r++) {
        if (ctx->ranks[r]->device < 0) continue;
        (void)hipSetDevice(ctx->ranks[r]->device);
        for (int b = 0; b < 2; b++) {
            if (r < m.d_sum[b].size() && m.d_sum[b][r]) (void)hipFree(m.d_sum[b][r]);
            if (r < m.d_sum[b].size()) m.d_sum[b][r] = nullptr;
        }
    }
    if (ctx->ranks[0]->device >= 0) {
        (void)hipSetDevice(ctx->ranks[0]->device);
        for (int b = 0; b < 2; b++) { if (m.d_total[b]) (void)hipFree(m.d_total[b]); m.d_total[b] = nullptr; }
    }
    m.sum_cap = 0;
}

// One frame over every GPU of the context, asynchronous: returns when every rank's work has been issued.  d_rgba8 /
// d_mean_rgb live on the first GPU.  The caller holds ctx->mu.
int multi_frame(rtr_context *ctx, const rtr_camera *camera, const rtr_settings *st, uint32_t width, uint32_t height, uint64_t seed,
                uint8_t *d_rgba8, float *d_mean_rgb) {
    MultiState &m = *ctx->multi;
    const uint32_t n = (uint32_t)ctx->ranks.size();
    if (ctx->ranks[0]->device < 0) return fail(ctx, RTR_ERR_NO_DEVICE, "host-only context: no HIP device bound (there is no CPU fallback)");
    if (!ctx->have_scene) return fail(ctx, RTR_ERR_NO_SCENE, "rtr_upload_scene has not been called");
    int rc = check_settings(ctx, camera, st, width, height);
    if (rc) return rc;
    const size_t bytes = (size_t)width * height * 3u * sizeof(float); // partial sums: three floats per pixel, packed
    if (m.sum_cap < bytes) { // (re)allocate the partial-sum buffers: waits for the frames in flight
        for (uint32_t r = 0; r < n; r++) { (void)hipSetDevice(ctx->ranks[r]->device); (void)hipDeviceSynchronize(); }
        multi_free_buffers(ctx);
        for (uint32_t r = 0; r < n; r++) {
            RTR_HIP_TRY(ctx, hipSetDevice(ctx->ranks[r]->device));
            for (int b = 0; b < 2; b++) RTR_HIP_TRY(ctx, hipMalloc(&m.d_sum[b][r], bytes));
        }
        RTR_HIP_TRY(ctx, hipSetDevice(ctx->ranks[0]->device));
        for (int b = 0; b < 2; b++) RTR_HIP_TRY(ctx, hipMalloc(&m.d_total[b], bytes));
        m.sum_cap = bytes;
        m.frames = 0;
    }
    const int buf = (int)(m.frames & 1u);
    const bool reuse = m.frames >= 2u; // the buffer's previous frame must have been reduced before it is overwritten
    const bool tiles = m.split_tiles;
    const rtr_camera cam = *camera;
    const rtr_settings settings = *st;
    if (m.same_device) { // test configuration: the ranks' buffer addresses for the adding kernel
        std::vector<const float *> ptrs(n);
        for (uint32_t r = 0; r < n; r++) ptrs[r] = static_cast<const float *>(m.d_sum[buf][r]);
        RTR_HIP_TRY(ctx, hipSetDevice(ctx->ranks[0]->device));
        if (!m.d_ptrs) RTR_HIP_TRY(ctx, hipMalloc(&m.d_ptrs, 64u * sizeof(void *)));
        RTR_HIP_TRY(ctx, hipMemcpyAsync(m.d_ptrs, ptrs.data(), n * sizeof(void *), hipMemcpyHostToDevice, m.comm_stream[0]));
        RTR_HIP_TRY(ctx, hipStreamSynchronize(m.comm_stream[0])); // (ptrs is a local)
    }
    for (uint32_t r = 0; r < n; r++) {
        rtr_context *rk = ctx->ranks[r];
        rk->tree_kernel = ctx->tree_kernel; rk->sample_chunks = ctx->sample_chunks; rk->chunk_taper = ctx->chunk_taper; rk->chunk_lengths = ctx->chunk_lengths; rk->block_samples = ctx->block_samples;
        rk->samples_batch = ctx->samples_batch; rk->stack_wide = ctx->stack_wide; rk->sky_bilinear = ctx->sky_bilinear;
        rk->chunk_timeout_s = ctx->chunk_timeout_s; rk->wf_flush = ctx->wf_flush; rk->debug_fault = ctx->debug_fault;
        rk->traversal_counters = ctx->traversal_counters;
        m.workers[r]->post([&m, ctx, rk, r, n, buf, reuse, tiles, cam, settings, width, height, seed, bytes, d_rgba8, d_mean_rgb] {
            std::lock_guard<std::recursive_mutex> g(rk->mu);
            int &out = m.rank_rc[r];
            out = RTR_OK;
            auto hip = [&](hipError_t e, const char *what) { if (e != hipSuccess && out == RTR_OK) out = fail_hip(rk, e, what); return e == hipSuccess; };
            if (!hip(hipSetDevice(rk->device), "hipSetDevice")) return;
            float *sum = static_cast<float *>(m.d_sum[buf][r]);
            if (reuse && !hip(hipStreamWaitEvent(rk->stream, m.ev_reduced[buf][r], 0), "hipStreamWaitEvent")) return;
            if (tiles) {
                if (!hip(hipMemsetAsync(sum, 0, bytes, rk->stream), "hipMemsetAsync")) return;
                out = launch_frame(rk, &cam, &settings, width, height, seed, r, n, sum, nullptr, nullptr, rk->stream);
            } else {
                uint32_t begin, count;
                multi_sample_range(r, n, settings.samples_per_pixel, begin, count);
                out = launch_frame(rk, &cam, &settings, width, height, seed, 0u, 1u, sum, nullptr, nullptr, rk->stream, begin, count);
            }
            if (out) return;
            if (!hip(hipEventRecord(m.ev_integrated[buf][r], rk->stream), "hipEventRecord")) return;
            hipStream_t cs = m.comm_stream[r];
            if (!m.same_device) {
                if (!hip(hipStreamWaitEvent(cs, m.ev_integrated[buf][r], 0), "hipStreamWaitEvent")) return;
                const ncclResult_t nr = m.rccl.Reduce(sum, r == 0 ? m.d_total[buf] : nullptr, (size_t)width * height * 3u, ncclFloat, ncclSum, 0,
                                                      m.comms[r], cs);
                if (nr != ncclSuccess) { out = fail(rk, RTR_ERR_HIP, std::string("ncclReduce: ") + m.rccl.GetErrorString(nr)); return; }
                if (!hip(hipEventRecord(m.ev_reduced[buf][r], cs), "hipEventRecord")) return;
            }
            (void)ctx; (void)d_rgba8; (void)d_mean_rgb;
        });
    }
    for (uint32_t r = 0; r < n; r++) m.workers[r]->wait();
    for (uint32_t r = 0; r < n; r++)
        if (m.rank_rc[r]) return multi_fail(ctx, ctx->ranks[r], m.rank_rc[r], r);
    // rank 0, communication stream: (same device: add the partial sums in rank order) then resolve
    rtr_context *root = ctx->ranks[0];
    RTR_HIP_TRY(ctx, hipSetDevice(root->device));
    hipStream_t cs = m.comm_stream[0];
    if (m.same_device) {
        for (uint32_t r = 0; r < n; r++) RTR_HIP_TRY(ctx, hipStreamWaitEvent(cs, m.ev_integrated[buf][r], 0));
        RTR_HIP_TRY(ctx, rtr::launch_sum_partials(static_cast<const float *const *>(m.d_ptrs), n, width * height, static_cast<float *>(m.d_total[buf]), cs));
        for (uint32_t r = 0; r < n; r++) RTR_HIP_TRY(ctx, hipEventRecord(m.ev_reduced[buf][r], cs));
    }
    RTR_HIP_TRY(ctx, rtr::launch_resolve(static_cast<const float *>(m.d_total[buf]), st->samples_per_pixel, width * height, d_rgba8, d_mean_rgb, cs));
    RTR_HIP_TRY(ctx, hipEventRecord(ctx->ev_res_stop, cs)); // (the multi context's own event: the frame is complete)
    m.frames++;
    ctx->launches++;
    return RTR_OK;
}

// waits for every GPU of the context and reports a rank's device fault
int multi_sync(rtr_context *ctx) {
    for (size_t r = 0; r < ctx->ranks.size(); r++) {
        rtr_context *rk = ctx->ranks[r];
        if (rk->device < 0) continue;
        RTR_HIP_TRY(ctx, hipSetDevice(rk->device));
        RTR_HIP_TRY(ctx, hipDeviceSynchronize());
    }
    int first = RTR_OK; // every rank's fault word is read (and cleared): one bad frame fails one call
    for (size_t r = 0; r < ctx->ranks.size(); r++) {
        rtr_context *rk = ctx->ranks[r];
        if (rk->device < 0 || rk->launches == 0) continue;
        const int rc = check_fault(rk);
        if (rc && !first) first = multi_fail(ctx, rk, rc, (uint32_t)r);
    }
    return first;
}

} // namespace

extern "C" {

int rtr_abi_version(void) { return RTR_ABI_VERSION; }

int rtr_create(int device, rtr_context **out) {
    if (!out) { g_create_error = "null out pointer"; return RTR_ERR_INVALID_ARGUMENT; }
    *out = nullptr;
    rtr_context *ctx = new (std::nothrow) rtr_context();
    if (!ctx) { g_create_error = "out of memory"; return RTR_ERR_OUT_OF_MEMORY; }
    if (device < 0) { // host-only
        ctx->device = -1;
        *out = ctx;
        return RTR_OK;
    }
    int count = 0;
    hipError_t e = hipGetDeviceCount(&count);
    if (e != hipSuccess || count <= 0 || device >= count) {
        g_create_error = std::string("no usable HIP device ") + std::to_string(device) + " (" +
                         (e != hipSuccess ? hipGetErrorString(e) : "device count too small") + "); there is no CPU fallback";
        delete ctx;
        return RTR_ERR_NO_DEVICE;
    }
    auto bail = [&](hipError_t err, const char *what) {
        g_create_error = std::string(what) + ": " + hipGetErrorString(err);
        rtr_destroy(ctx);
        return (int)RTR_ERR_HIP;
    };
    ctx->device = device;
    if ((e = hipSetDevice(device)) != hipSuccess) { ctx->device = -1; return bail(e, "hipSetDevice"); }
    if ((e = hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking)) != hipSuccess) return bail(e, "hipStreamCreate");
    for (int i = 0; i < kRing; i++) {
        if ((e = hipEventCreate(&ctx->ev_start[i])) != hipSuccess) return bail(e, "hipEventCreate");
        if ((e = hipEventCreate(&ctx->ev_stop[i])) != hipSuccess) return bail(e, "hipEventCreate");
        if ((e = hipEventCreate(&ctx->ev_mid[i])) != hipSuccess) return bail(e, "hipEventCreate");
    }
    if ((e = hipEventCreate(&ctx->ev_res_start)) != hipSuccess) return bail(e, "hipEventCreate");
    if ((e = hipEventCreate(&ctx->ev_res_stop)) != hipSuccess) return bail(e, "hipEventCreate");
    // kRing segment counters and the sticky fault word
    if ((e = hipMalloc((void **)&ctx->d_counters, (kRing + 1) * sizeof(unsigned long long))) != hipSuccess) return bail(e, "hipMalloc");
    if ((e = hipMemset(ctx->d_counters, 0, (kRing + 1) * sizeof(unsigned long long))) != hipSuccess) return bail(e, "hipMemset");
    if ((e = hipMalloc((void **)&ctx->d_work, 256)) != hipSuccess) return bail(e, "hipMalloc");
    *out = ctx;
    return RTR_OK;
}

int rtr_sample_range(uint32_t rank, uint32_t n_ranks, uint32_t samples_per_pixel, uint32_t *begin, uint32_t *count) {
    if (!begin || !count || n_ranks == 0u || rank >= n_ranks) return RTR_ERR_INVALID_ARGUMENT;
    multi_sample_range(rank, n_ranks, samples_per_pixel, *begin, *count);
    return RTR_OK;
}

int rtr_create_multi(const int *device_ids, int n_devices, rtr_context **out) {
    if (!out) { g_create_error = "null out pointer"; return RTR_ERR_INVALID_ARGUMENT; }
    *out = nullptr;
    if (!device_ids || n_devices < 1 || n_devices > 64) { g_create_error = "rtr_create_multi: 1 to 64 devices"; return RTR_ERR_INVALID_ARGUMENT; }
    const uint32_t n = (uint32_t)n_devices;
    bool all_same = true, all_distinct = true, any_host = false, all_host = true;
    for (uint32_t i = 0; i < n; i++) {
        any_host |= device_ids[i] < 0; all_host &= device_ids[i] < 0;
        for (uint32_t j = 0; j < i; j++) { if (device_ids[i] == device_ids[j]) all_distinct = false; else all_same = false; }
    }
    if (any_host && !all_host) { g_create_error = "rtr_create_multi: host-only (-1) and GPU ranks cannot be mixed"; return RTR_ERR_INVALID_ARGUMENT; }
    if (!all_host && !all_distinct && !all_same) {
        g_create_error = "rtr_create_multi: device ids must be all distinct (or all the same device: a test configuration)";
        return RTR_ERR_INVALID_ARGUMENT;
    }
    rtr_context *ctx = new (std::nothrow) rtr_context();
    if (!ctx) { g_create_error = "out of memory"; return RTR_ERR_OUT_OF_MEMORY; }
    ctx->multi = new (std::nothrow) MultiState();
    if (!ctx->multi) { delete ctx; g_create_error = "out of memory"; return RTR_ERR_OUT_OF_MEMORY; }
    MultiState &m = *ctx->multi;
    auto bail = [&](int code, const std::string &msg) {
        g_create_error = msg;
        rtr_destroy(ctx);
        return code;
    };
    for (uint32_t r = 0; r < n; r++) {
        rtr_context *rk = nullptr;
        const int rc = rtr_create(device_ids[r], &rk);
        if (rc != RTR_OK) return bail(rc, g_create_error);
        ctx->ranks.push_back(rk);
    }
    ctx->device = ctx->ranks[0]->device;
    m.same_device = !all_host && n > 1u && all_same;
    m.rank_rc.assign(n, RTR_OK);
    for (int b = 0; b < 2; b++) { m.d_sum[b].assign(n, nullptr); m.ev_integrated[b].assign(n, nullptr); m.ev_reduced[b].assign(n, nullptr); }
    m.comm_stream.assign(n, nullptr);
    for (uint32_t r = 0; r < n; r++) m.workers.emplace_back(new RankWorker());
    if (all_host) { *out = ctx; return RTR_OK; } // host-only: scene lowering and introspection; every compute call fails
    for (uint32_t r = 0; r < n; r++) {
        hipError_t e = hipSetDevice(device_ids[r]);
        if (e == hipSuccess) e = hipStreamCreateWithFlags(&m.comm_stream[r], hipStreamNonBlocking);
        for (int b = 0; b < 2 && e == hipSuccess; b++) {
            e = hipEventCreateWithFlags(&m.ev_integrated[b][r], hipEventDisableTiming);
            if (e == hipSuccess) e = hipEventCreateWithFlags(&m.ev_reduced[b][r], hipEventDisableTiming);
        }
        if (e != hipSuccess) return bail(RTR_ERR_HIP, std::string("rtr_create_multi: ") + hipGetErrorString(e));
    }
    {
        hipError_t e = hipSetDevice(ctx->device);
        if (e == hipSuccess) e = hipEventCreate(&ctx->ev_res_stop);
        if (e != hipSuccess) return bail(RTR_ERR_HIP, std::string("rtr_create_multi: ") + hipGetErrorString(e));
    }
    if (!m.same_device) { // one communicator per GPU, all created by this process (ncclCommInitAll)
        std::string err;
        if (!load_rccl(m.rccl, err)) return bail(RTR_ERR_UNSUPPORTED, err);
        m.comms.assign(n, nullptr);
        const ncclResult_t nr = m.rccl.CommInitAll(m.comms.data(), (int)n, device_ids);
        if (nr != ncclSuccess) { m.comms.clear(); return bail(RTR_ERR_HIP, std::string("ncclCommInitAll: ") + m.rccl.GetErrorString(nr)); }
    }
    *out = ctx;
    return RTR_OK;
}

void rtr_destroy(rtr_context *ctx) {
    if (!ctx) return;
    if (ctx->multi) {
        MultiState &m = *ctx->multi;
        for (rtr_context *rk : ctx->ranks)
            if (rk->device >= 0) { (void)hipSetDevice(rk->device); (void)hipDeviceSynchronize(); }
        m.workers.clear(); // joins the rank threads
        for (ncclComm_t c : m.comms) if (c) (void)m.rccl.CommDestroy(c);
        if (!ctx->ranks.empty()) multi_free_buffers(ctx);
        for (size_t r = 0; r < ctx->ranks.size(); r++) {
            if (ctx->ranks[r]->device < 0) continue;
            (void)hipSetDevice(ctx->ranks[r]->device);
            if (r < m.comm_stream.size() && m.comm_stream[r]) (void)hipStreamDestroy(m.comm_stream[r]);
            for (int b = 0; b < 2; b++) {
                if (r < m.ev_integrated[b].size() && m.ev_integrated[b][r]) (void)hipEventDestroy(m.ev_integrated[b][r]);
                if (r < m.ev_reduced[b].size() && m.ev_reduced[b][r]) (void)hipEventDestroy(m.ev_reduced[b][r]);
            }
        }
        if (ctx->device >= 0) {
            (void)hipSetDevice(ctx->device);
            if (m.d_ptrs) (void)hipFree(m.d_ptrs);
            if (ctx->ev_res_stop) (void)hipEventDestroy(ctx->ev_res_stop);
            if (ctx->d_rgba8) (void)hipFree(ctx->d_rgba8);
            if (ctx->d_mean) (void)hipFree(ctx->d_mean);
        }
        for (rtr_context *rk : ctx->ranks) rtr_destroy(rk);
        delete ctx->multi;
        delete ctx;
        return;
    }
    if (ctx->device >= 0) {
        (void)hipSetDevice(ctx->device);
        if (ctx->stream) (void)hipStreamSynchronize(ctx->stream);
        free_scene(ctx);
        for (int i = 0; i < kRing; i++) {
            if (ctx->ev_start[i]) (void)hipEventDestroy(ctx->ev_start[i]);
            if (ctx->ev_stop[i]) (void)hipEventDestroy(ctx->ev_stop[i]);
            if (ctx->ev_mid[i]) (void)hipEventDestroy(ctx->ev_mid[i]);
        }
        if (ctx->ev_res_start) (void)hipEventDestroy(ctx->ev_res_start);
        if (ctx->ev_res_stop) (void)hipEventDestroy(ctx->ev_res_stop);
        if (ctx->d_counters) (void)hipFree(ctx->d_counters);
        if (ctx->d_carry) (void)hipFree(ctx->d_carry);
        if (ctx->d_samples) (void)hipFree(ctx->d_samples);
        if (ctx->d_wf) (void)hipFree(ctx->d_wf);
        if (ctx->d_work) (void)hipFree(ctx->d_work);
        if (ctx->d_chunk_flags) (void)hipFree(ctx->d_chunk_flags);
        if (ctx->d_rgba8) (void)hipFree(ctx->d_rgba8);
        if (ctx->d_mean) (void)hipFree(ctx->d_mean);
        if (ctx->stream) (void)hipStreamDestroy(ctx->stream);
    }
    delete ctx;
}

const char *rtr_last_error(const rtr_context *ctx) { return ctx ? ctx->error.c_str() : g_create_error.c_str(); }

int rtr_upload_scene(rtr_context *ctx, const rtr_scene_desc *scene) {
    if (!ctx) return RTR_ERR_INVALID_ARGUMENT;
    RTR_LOCK(ctx);
    if (!scene) return fail(ctx, RTR_ERR_INVALID_ARGUMENT, "null scene");
    auto t0 = std::chrono::steady_clock::now();
    rtr::FlatScene flat;
    std::string err;
    int rc = rtr::build_flat_scene(*scene, flat, err, ctx->build_options);
    if (rc != RTR_OK) return fail(ctx, rc, err);
    if (!ctx->ranks.empty()) return multi_upload(ctx, flat, t0); // replicated on every GPU of the context
    return install_scene(ctx, flat, t0);
}

int rtr_set_option(rtr_context *ctx, const char *name, const char *value) {
    if (!ctx || !name || !value) return RTR_ERR_INVALID_ARGUMENT;
    RTR_LOCK(ctx);
    const std::string n(name), v(value);
    if (n == "mesh_tree") { // takes effect at the next rtr_upload_scene
        if (v == "reference") ctx->build_options.sah_mesh_trees = false;
        else if (v == "sah") { ctx->build_options.sah_mesh_trees = true; ctx->build_options.lbvh_mesh_trees = false; return RTR_OK; }
        else if (v == "lbvh") { ctx->build_options.lbvh_mesh_trees = true; ctx->build_options.sah_mesh_trees = false; return RTR_OK; }
        else return fail(ctx, RTR_ERR_INVALID_ARGUMENT, "mesh_tree must be \"reference\", \"sah\" or \"lbvh\"");
        ctx->build_options.lbvh_mesh_trees = false;
        return RTR_OK;
    }
    if (n == "scene_tree") {
        if (v == "reference") ctx->build_options.sah_scene_tree = false;
        else if (v == "sah") ctx->build_options.sah_scene_tree = true;
        else return fail(ctx, RTR_ERR_INVALID_ARGUMENT, "scene_tree must be \"reference\" or \"sah\"");
        return RTR_OK;
    }
    if (n == "tree_kernel") { // takes effect at the next render
        if (v == "auto") ctx->tree_kernel = 0;
        else if (v == "resumable") ctx->tree_kernel = 1;
        else if (v == "wave") ctx->tree_kernel = 2;
        else if (v == "samples") ctx->tree_kernel = 3;
        else if (v == "stream") ctx->tree_kernel = 4;
        else return fail(ctx, RTR_ERR_INVALID_ARGUMENT, "tree_kernel must be \"auto\", \"stream\", \"samples\", \"resumable\" or \"wave\"");
        return RTR_OK;
    }
    if (n == "stream_flush") { // takes effect at the next render
        char *end = nullptr;
        const unsigned long k = std::strtoul(v.c_str(), &end, 10);
        if (v.empty() || *end != '\0' || k < 1ul || k > 64ul) return fail(ctx, RTR_ERR_INVALID_ARGUMENT, "stream_flush must be an integer in [1, 64]");
        ctx->wf_flush = (uint32_t)k;
        return RTR_OK;
    }
    if (n == "samples_batch") { // takes effect at the next render; tests use it to force several batches
        char *end = nullptr;
        const unsigned long k = std::strtoul(v.c_str(), &end, 10);
        if (v.empty() || *end != '\0' || k > 0xFFFFFFFFul) return fail(ctx, RTR_ERR_INVALID_ARGUMENT, "samples_batch must be an integer >= 0 (0 = automatic)");
        ctx->samples_batch = (uint32_t)k;
        return RTR_OK;
    }
    if (n == "sample_chunks") { // takes effect at the next render (lengths given through "debug_chunk_lengths" go with their count)
        if (v == "auto") { ctx->sample_chunks = 0; ctx->chunk_lengths.clear(); return RTR_OK; }
        char *end = nullptr;
        const unsigned long k = std::strtoul(v.c_str(), &end, 10);
        if (v.empty() || *end != '\0' || k < 1ul || k > 4096ul) return fail(ctx, RTR_ERR_INVALID_ARGUMENT, "sample_chunks must be \"auto\" or an integer in [1, 4096]");
        ctx->sample_chunks = (int)k;
        ctx->chunk_lengths.clear();
        return RTR_OK;
    }
    if (n == "debug_chunk_lengths") { // measuring aid: relative chunk lengths, e.g. "24:7:1" (any values give the same frame)
        std::vector<uint32_t> lens;
        size_t pos = 0;
        while (pos <= v.size()) {
            const size_t colon = std::min(v.find(':', pos), v.size());
            char *end = nullptr;
            const std::string item = v.substr(pos, colon - pos);
            const unsigned long k = std::strtoul(item.c_str(), &end, 10);
            if (item.empty() || *end != '\0' || k < 1ul || k > 65536ul) return fail(ctx, RTR_ERR_INVALID_ARGUMENT, "debug_chunk_lengths must be 1..8 positive integers separated by ':'");
            lens.push_back((uint32_t)k);
            pos = colon + 1;
        }
        if (lens.empty() || lens.size() > 8u) return fail(ctx, RTR_ERR_INVALID_ARGUMENT, "debug_chunk_lengths must be 1..8 positive integers separated by ':'");
        ctx->chunk_lengths = lens;
        ctx->sample_chunks = (int)lens.size();
        return RTR_OK;
    }
    if (n == "debug_chunk_taper") { // measuring aid: how steeply the chunk lengths fall (any value gives the same frame)
        if (v == "auto") { ctx->chunk_taper = -1; return RTR_OK; }
        if (v.size() != 1 || v[0] < '0' || v[0] > '7') return fail(ctx, RTR_ERR_INVALID_ARGUMENT, "debug_chunk_taper must be \"auto\" or 0..7");
        ctx->chunk_taper = v[0] - '0';
        return RTR_OK;
    }
    if (n == "chunk_timeout_ms") { // takes effect at the next render
        char *end = nullptr;
        const double ms = std::strtod(v.c_str(), &end);
        if (v.empty() || *end != '\0' || !(ms >= 0.0) || ms > 6.0e5) return fail(ctx, RTR_ERR_INVALID_ARGUMENT, "chunk_timeout_ms must be a number in [0, 600000]");
        ctx->chunk_timeout_s = ms * 1.0e-3;
        return RTR_OK;
    }
    if (n == "sky_filter") { // takes effect at the next render
        if (v == "nearest") ctx->sky_bilinear = false;
        else if (v == "bilinear") ctx->sky_bilinear = true;
        else return fail(ctx, RTR_ERR_INVALID_ARGUMENT, "sky_filter must be \"nearest\" or \"bilinear\"");
        return RTR_OK;
    }
    if (n == "debug_alloc_limit") { // testing aid: scratch buffers above this size cannot be allocated (0 = no limit)
        char *end = nullptr;
        const unsigned long long k = std::strtoull(v.c_str(), &end, 10);
        if (v.empty() || *end != '\0') return fail(ctx, RTR_ERR_INVALID_ARGUMENT, "debug_alloc_limit must be a byte count");
        ctx->debug_alloc_limit = k;
        return RTR_OK;
    }
    if (n == "block_samples") { // takes effect at the next render
        char *end = nullptr;
        const unsigned long k = std::strtoul(v.c_str(), &end, 10);
        if (v.empty() || *end != '\0' || k < 1ul || k > 1024ul) return fail(ctx, RTR_ERR_INVALID_ARGUMENT, "block_samples must be an integer in [1, 1024]");
        ctx->block_samples = (uint32_t)k;
        return RTR_OK;
    }
    if (n == "stack_entries") { // takes effect at the next render
        if (v == "auto") ctx->stack_wide = false;
        else if (v == "wide") ctx->stack_wide = true;
        else return fail(ctx, RTR_ERR_INVALID_ARGUMENT, "stack_entries must be \"auto\" or \"wide\"");
        return RTR_OK;
    }
    if (n == "debug_lds_pad") { // measuring aid: extra dynamic LDS bytes per workgroup (occupancy experiments)
        char *end = nullptr;
        const unsigned long k = std::strtoul(v.c_str(), &end, 10);
        if (v.empty() || *end != '\0' || k > 65536ul) return fail(ctx, RTR_ERR_INVALID_ARGUMENT, "debug_lds_pad must be a byte count up to 65536");
        ctx->debug_lds_pad = (uint32_t)k;
        return RTR_OK;
    }
    if (n == "debug_fault") { // testing aid: the following launches set the fault word
        if (v == "on") ctx->debug_fault = true;
        else if (v == "off") ctx->debug_fault = false;
        else return fail(ctx, RTR_ERR_INVALID_ARGUMENT, "debug_fault must be \"on\" or \"off\"");
        return RTR_OK;
    }
    if (n == "multi_split") { // rtr_create_multi contexts: how a frame is divided among the GPUs (applied at the next render)
        if (!ctx->multi) return fail(ctx, RTR_ERR_INVALID_ARGUMENT, "multi_split applies to contexts made by rtr_create_multi");
        if (v == "samples") ctx->multi->split_tiles = false;
        else if (v == "tiles") ctx->multi->split_tiles = true;
        else return fail(ctx, RTR_ERR_INVALID_ARGUMENT, "multi_split must be \"samples\" or \"tiles\"");
        return RTR_OK;
    }
    if (n == "traversal_counters") { // takes effect at the next render
        if (v == "on") ctx->traversal_counters = true;
        else if (v == "off") ctx->traversal_counters = false;
        else return fail(ctx, RTR_ERR_INVALID_ARGUMENT, "traversal_counters must be \"on\" or \"off\"");
        return RTR_OK;
    }
    if (n == "leaf_pairs") {
        if (v == "on") ctx->build_options.leaf_pairs = true;
        else if (v == "off") ctx->build_options.leaf_pairs = false;
        else return fail(ctx, RTR_ERR_INVALID_ARGUMENT, "leaf_pairs must be \"on\" or \"off\"");
        return RTR_OK;
    }
    return fail(ctx, RTR_ERR_INVALID_ARGUMENT, "unknown option " + n);
}

int rtr_scene_info_get(const rtr_context *ctx, rtr_scene_info *out) {
    if (!ctx || !out) return RTR_ERR_INVALID_ARGUMENT;
    if (!ctx->ranks.empty()) ctx = ctx->ranks[0]; // every GPU holds the same scene
    RTR_LOCK(ctx);
    if (!ctx->have_scene) return RTR_ERR_NO_SCENE;
    const rtr::FlatScene &fs = ctx->flat;
    std::memset(out, 0, sizeof *out);
    out->n_objects = fs.n_objects; out->n_spheres = fs.n_spheres; out->n_meshes = fs.n_meshes;
    out->n_triangles = fs.n_triangles; out->n_materials = (uint32_t)fs.materials.size();
    out->n_leaves = (uint32_t)fs.leaf_hot.size(); out->n_nodes = fs.split_nodes;
    out->tree_height = fs.tree_height; out->spheres_only = fs.spheres_only ? 1u : 0u;
    out->device_bytes = ctx->device_bytes;
    return RTR_OK;
}

int64_t rtr_bvh_leaf_order(const rtr_context *ctx, uint64_t *out, uint64_t cap) {
    if (!ctx) return RTR_ERR_INVALID_ARGUMENT;
    if (!ctx->ranks.empty()) ctx = ctx->ranks[0];
    RTR_LOCK(ctx);
    if (!ctx->have_scene) return RTR_ERR_NO_SCENE;
    const auto &cold = ctx->flat.leaf_cold;
    if (out)
        for (uint64_t i = 0; i < cold.size() && i < cap; i++) out[i] = ((uint64_t)cold[i].object << 32) | cold[i].handle;
    return (int64_t)cold.size();
}

int rtr_render_frame_device(rtr_context *ctx, const rtr_camera *camera, const rtr_settings *settings, uint32_t width,
                            uint32_t height, uint64_t seed, uint8_t *d_rgba8, void *stream) {
    if (!ctx) return RTR_ERR_INVALID_ARGUMENT;
    RTR_LOCK(ctx);
    if (!d_rgba8) return fail(ctx, RTR_ERR_INVALID_ARGUMENT, "null output");
    if (ctx->multi) return multi_frame(ctx, camera, settings, width, height, seed, d_rgba8, nullptr); // (its own streams)
    return launch_frame(ctx, camera, settings, width, height, seed, 0u, 1u, nullptr, d_rgba8, nullptr, (hipStream_t)stream);
}

int rtr_render_tiles_device(rtr_context *ctx, const rtr_camera *camera, const rtr_settings *settings, uint32_t width,
                            uint32_t height, uint64_t seed, uint32_t shard, uint32_t n_shards, float *d_sum_rgba,
                            void *stream) {
    if (!ctx) return RTR_ERR_INVALID_ARGUMENT;
    RTR_LOCK(ctx);
    if (!d_sum_rgba) return fail(ctx, RTR_ERR_INVALID_ARGUMENT, "null output");
    if (ctx->multi) return fail(ctx, RTR_ERR_UNSUPPORTED, "a multi-GPU context shards its frames itself: use rtr_render* / rtr_render_frame_device");
    return launch_frame(ctx, camera, settings, width, height, seed, shard, n_shards, d_sum_rgba, nullptr, nullptr, (hipStream_t)stream);
}

int rtr_render_samples_device(rtr_context *ctx, const rtr_camera *camera, const rtr_settings *settings, uint32_t width,
                              uint32_t height, uint64_t seed, uint32_t sample_begin, uint32_t sample_count,
                              float *d_sum_rgba, void *stream) {
    if (!ctx) return RTR_ERR_INVALID_ARGUMENT;
    RTR_LOCK(ctx);
    if (!d_sum_rgba) return fail(ctx, RTR_ERR_INVALID_ARGUMENT, "null output");
    if (ctx->multi) return fail(ctx, RTR_ERR_UNSUPPORTED, "a multi-GPU context shards its frames itself: use rtr_render* / rtr_render_frame_device");
    return launch_frame(ctx, camera, settings, width, height, seed, 0u, 1u, d_sum_rgba, nullptr, nullptr, (hipStream_t)stream,
                        sample_begin, sample_count);
}

int rtr_resolve_device(rtr_context *ctx, const float *d_sum_rgba, uint32_t samples_per_pixel, uint32_t width,
                       uint32_t height, uint8_t *d_rgba8, float *d_mean_rgb, void *stream) {
    if (!ctx) return RTR_ERR_INVALID_ARGUMENT;
    RTR_LOCK(ctx);
    if (ctx->device < 0) return fail(ctx, RTR_ERR_NO_DEVICE, "host-only context: no HIP device bound (there is no CPU fallback)");
    if (!d_sum_rgba || width == 0 || height == 0) return fail(ctx, RTR_ERR_INVALID_ARGUMENT, "bad resolve arguments");
    if (ctx->multi) ctx = ctx->ranks[0]; // (rank 0's device is where a multi-GPU context resolves)
    RTR_HIP_TRY(ctx, hipSetDevice(ctx->device));
    hipStream_t s = stream ? (hipStream_t)stream : ctx->stream;
    RTR_HIP_TRY(ctx, hipEventRecord(ctx->ev_res_start, s));
    RTR_HIP_TRY(ctx, rtr::launch_resolve(d_sum_rgba, samples_per_pixel, width * height, d_rgba8, d_mean_rgb, s));
    RTR_HIP_TRY(ctx, hipEventRecord(ctx->ev_res_stop, s));
    ctx->resolve_timed = true;
    return RTR_OK;
}

int rtr_render(rtr_context *ctx, const rtr_camera *camera, const rtr_settings *settings, uint32_t width, uint32_t height,
               uint64_t seed, uint8_t *rgba8, float *mean_rgb) {
    if (!ctx) return RTR_ERR_INVALID_ARGUMENT;
    RTR_LOCK(ctx);
    if (!rgba8 && !mean_rgb) return fail(ctx, RTR_ERR_INVALID_ARGUMENT, "no output requested");
    if (ctx->device < 0) return fail(ctx, RTR_ERR_NO_DEVICE, "host-only context: no HIP device bound (there is no CPU fallback)");
    int rc = check_settings(ctx, camera, settings, width, height);
    if (rc) return rc;
    RTR_HIP_TRY(ctx, hipSetDevice(ctx->device));
    size_t npx = (size_t)width * height;
    uint8_t *d8 = nullptr;
    float *dm = nullptr;
    if (rgba8) {
        if ((rc = ensure_scratch(ctx, &ctx->d_rgba8, &ctx->rgba8_cap, npx * 4))) return rc;
        d8 = static_cast<uint8_t *>(ctx->d_rgba8);
    }
    if (mean_rgb) {
        if ((rc = ensure_scratch(ctx, &ctx->d_mean, &ctx->mean_cap, npx * 3 * sizeof(float)))) return rc;
        dm = static_cast<float *>(ctx->d_mean);
    }
    if (ctx->multi) { // every GPU of the context; reduced and resolved on the first one's communication stream
        if ((rc = multi_frame(ctx, camera, settings, width, height, seed, d8, dm))) return rc;
        hipStream_t cs = ctx->multi->comm_stream[0];
        RTR_HIP_TRY(ctx, hipSetDevice(ctx->device));
        if (rgba8) RTR_HIP_TRY(ctx, hipMemcpyAsync(rgba8, d8, npx * 4, hipMemcpyDeviceToHost, cs));
        if (mean_rgb) RTR_HIP_TRY(ctx, hipMemcpyAsync(mean_rgb, dm, npx * 3 * sizeof(float), hipMemcpyDeviceToHost, cs));
        RTR_HIP_TRY(ctx, hipStreamSynchronize(cs));
        return multi_sync(ctx);
    }
    rc = launch_frame(ctx, camera, settings, width, height, seed, 0u, 1u, nullptr, d8, dm, ctx->stream);
    if (rc) return rc;
    if (rgba8) RTR_HIP_TRY(ctx, hipMemcpyAsync(rgba8, d8, npx * 4, hipMemcpyDeviceToHost, ctx->stream));
    if (mean_rgb) RTR_HIP_TRY(ctx, hipMemcpyAsync(mean_rgb, dm, npx * 3 * sizeof(float), hipMemcpyDeviceToHost, ctx->stream));
    RTR_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return check_fault(ctx);
}

int rtr_render_rgba8(rtr_context *ctx, const rtr_camera *camera, const rtr_settings *settings, uint32_t width,
                     uint32_t height, uint64_t seed, uint8_t *rgba8) {
    if (!ctx) return RTR_ERR_INVALID_ARGUMENT;
    if (!rgba8) return fail(ctx, RTR_ERR_INVALID_ARGUMENT, "null output");
    return rtr_render(ctx, camera, settings, width, height, seed, rgba8, nullptr);
}

int rtr_render_f32(rtr_context *ctx, const rtr_camera *camera, const rtr_settings *settings, uint32_t width,
                   uint32_t height, uint64_t seed, float *mean_rgb) {
    if (!ctx) return RTR_ERR_INVALID_ARGUMENT;
    if (!mean_rgb) return fail(ctx, RTR_ERR_INVALID_ARGUMENT, "null output");
    return rtr_render(ctx, camera, settings, width, height, seed, nullptr, mean_rgb);
}

int rtr_get_stats(rtr_context *ctx, rtr_stats *out) {
    if (!ctx || !out) return RTR_ERR_INVALID_ARGUMENT;
    RTR_LOCK(ctx);
    std::memset(out, 0, sizeof *out);
    out->upload_ms = ctx->upload_ms;
    out->launches = ctx->launches;
    out->primary_samples = ctx->last_primary;
    out->lds_bytes = ctx->last_lds;
    out->vgprs = ctx->last_vgprs;
    out->block_threads = 64;
    if (ctx->multi) { // the frame's totals over the GPUs: samples and segments add up, the time is the slowest GPU's integrate
        if (ctx->device < 0 || ctx->launches == 0) return RTR_OK;
        if (int rc = multi_sync(ctx)) return rc;
        for (rtr_context *rk : ctx->ranks) {
            rtr_stats s;
            const int rc = rtr_get_stats(rk, &s);
            if (rc) return fail(ctx, rc, rk->error);
            out->primary_samples += s.primary_samples;
            out->segments += s.segments;
            out->kernel_ms = std::max(out->kernel_ms, s.kernel_ms);
            out->vgprs = s.vgprs; out->lds_bytes = s.lds_bytes;
            out->node_steps += s.node_steps; out->leaf_tests += s.leaf_tests; out->sky_lookups += s.sky_lookups;
            out->traversal_counted = s.traversal_counted;
        }
        return RTR_OK;
    }
    if (ctx->device < 0 || ctx->launches == 0) return RTR_OK;
    RTR_HIP_TRY(ctx, hipSetDevice(ctx->device));
    RTR_HIP_TRY(ctx, hipDeviceSynchronize());
    if (int rc = check_fault(ctx)) return rc;
    uint32_t slot = (ctx->launches - 1u) % kRing;
    float ms = 0.0f;
    RTR_HIP_TRY(ctx, hipEventElapsedTime(&ms, ctx->ev_start[slot], ctx->ev_stop[slot]));
    out->kernel_ms = ms;
    if (ctx->resolve_timed) {
        RTR_HIP_TRY(ctx, hipEventElapsedTime(&ms, ctx->ev_res_start, ctx->ev_res_stop));
        out->resolve_ms = ms;
    }
    unsigned long long seg = 0;
    RTR_HIP_TRY(ctx, hipMemcpy(&seg, ctx->d_counters + slot, sizeof seg, hipMemcpyDeviceToHost));
    out->segments = seg;
    if (ctx->last_counted) {
        unsigned long long w[3] = {0, 0, 0};
        RTR_HIP_TRY(ctx, hipMemcpy(w, reinterpret_cast<unsigned long long *>(ctx->d_work) + 1, sizeof w, hipMemcpyDeviceToHost));
        out->node_steps = w[0]; out->leaf_tests = w[1]; out->sky_lookups = w[2];
        out->traversal_counted = 1u;
    }
    return RTR_OK;
}

int rtr_kernel_times_ms(rtr_context *ctx, float *out, uint32_t cap) {
    if (!ctx || !out) return RTR_ERR_INVALID_ARGUMENT;
    RTR_LOCK(ctx);
    if (ctx->multi) { // per frame: the slowest GPU's time
        int n = 0;
        std::vector<float> tmp(cap ? cap : 1u);
        for (size_t r = 0; r < ctx->ranks.size(); r++) {
            const int k = rtr_kernel_times_ms(ctx->ranks[r], tmp.data(), cap);
            if (k < 0) return fail(ctx, k, ctx->ranks[r]->error);
            if (r == 0) { n = k; std::copy(tmp.begin(), tmp.begin() + k, out); }
            else for (int i = 0; i < std::min(n, k); i++) out[n - 1 - i] = std::max(out[n - 1 - i], tmp[(size_t)(k - 1 - i)]);
        }
        return n;
    }
    if (ctx->device < 0 || ctx->launches == 0) return 0;
    RTR_HIP_TRY(ctx, hipSetDevice(ctx->device));
    RTR_HIP_TRY(ctx, hipDeviceSynchronize());
    if (int rc = check_fault(ctx)) return rc;
    uint32_t n = ctx->launches < (uint32_t)kRing ? ctx->launches : (uint32_t)kRing;
    if (n > cap) n = cap;
    for (uint32_t i = 0; i < n; i++) {
        uint32_t slot = (ctx->launches - n + i) % kRing;
        float ms = 0.0f;
        RTR_HIP_TRY(ctx, hipEventElapsedTime(&ms, ctx->ev_start[slot], ctx->ev_stop[slot]));
        out[i] = ms;
    }
    return (int)n;
}

int rtr_integrate_times_ms(rtr_context *ctx, float *out, uint32_t cap) {
    if (!ctx || !out) return RTR_ERR_INVALID_ARGUMENT;
    RTR_LOCK(ctx);
    if (ctx->multi) { // per frame: the slowest GPU's time
        int n = 0;
        std::vector<float> tmp(cap ? cap : 1u);
        for (size_t r = 0; r < ctx->ranks.size(); r++) {
            const int k = rtr_integrate_times_ms(ctx->ranks[r], tmp.data(), cap);
            if (k < 0) return fail(ctx, k, ctx->ranks[r]->error);
            if (r == 0) { n = k; std::copy(tmp.begin(), tmp.begin() + k, out); }
            else for (int i = 0; i < std::min(n, k); i++) out[n - 1 - i] = std::max(out[n - 1 - i], tmp[(size_t)(k - 1 - i)]);
        }
        return n;
    }
    if (ctx->device < 0 || ctx->launches == 0) return 0;
    RTR_HIP_TRY(ctx, hipSetDevice(ctx->device));
    RTR_HIP_TRY(ctx, hipDeviceSynchronize());
    if (int rc = check_fault(ctx)) return rc;
    uint32_t n = ctx->launches < (uint32_t)kRing ? ctx->launches : (uint32_t)kRing;
    if (n > cap) n = cap;
    for (uint32_t i = 0; i < n; i++) {
        uint32_t slot = (ctx->launches - n + i) % kRing;
        float ms = 0.0f;
        RTR_HIP_TRY(ctx, hipEventElapsedTime(&ms, ctx->ev_start[slot], ctx->ev_mid[slot]));
        out[i] = ms;
    }
    return (int)n;
}

const char *rtr_kernel_name(const rtr_context *ctx) {
    if (!ctx) return "";
    if (!ctx->ranks.empty()) ctx = ctx->ranks[0];
    RTR_LOCK(ctx);
    thread_local std::string copy; // the caller's own copy: a render on another thread cannot invalidate it
    copy = ctx->kernel_name;
    return copy.c_str();
}

int rtr_camera_ray_origin(const rtr_camera *camera, float out[12]) {
    if (!camera || !out) return RTR_ERR_INVALID_ARGUMENT;
    rtr::camera_ray_origin(*camera, out, out + 3, out + 6, out + 9);
    return RTR_OK;
}

int rtr_probe_hits(rtr_context *ctx, uint32_t enable_bvh_tree, uint64_t seed, uint32_t n_rays, const float *origins, const float *directions,
                   rtr_hit *out) {
    if (!ctx) return RTR_ERR_INVALID_ARGUMENT;
    if (!ctx->ranks.empty()) ctx = ctx->ranks[0];
    RTR_LOCK(ctx);
    static_assert(sizeof(rtr_hit) == 48, "k_probe_hits writes 12 words per ray");
    if (ctx->device < 0) return fail(ctx, RTR_ERR_NO_DEVICE, "host-only context: no HIP device bound (there is no CPU fallback)");
    if (!ctx->have_scene) return fail(ctx, RTR_ERR_NO_SCENE, "rtr_upload_scene has not been called");
    if (n_rays == 0u) return RTR_OK;
    if (!origins || !directions || !out || n_rays > (1u << 24)) return fail(ctx, RTR_ERR_INVALID_ARGUMENT, "bad probe arguments");
    RTR_HIP_TRY(ctx, hipSetDevice(ctx->device));
    rtr::RenderParams p;
    std::memset(&p, 0, sizeof p);
    set_world_params(ctx, enable_bvh_tree != 0u, p);
    seed_keys(seed, p.seed_key0, p.seed_key1);
    if (p.stack_cap * 64u * 6u > 64u * 1024u) return fail(ctx, RTR_ERR_UNSUPPORTED, "tree height exceeds the probe kernel's LDS");
    const size_t in_bytes = (size_t)n_rays * 3u * sizeof(float), out_bytes = (size_t)n_rays * sizeof(rtr_hit);
    void *d = nullptr;
    RTR_HIP_TRY(ctx, hipMalloc(&d, 2u * in_bytes + out_bytes));
    float *d_o = static_cast<float *>(d), *d_d = d_o + (size_t)n_rays * 3u;
    uint32_t *d_out = reinterpret_cast<uint32_t *>(d_d + (size_t)n_rays * 3u);
    hipError_t e = hipMemcpyAsync(d_o, origins, in_bytes, hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(d_d, directions, in_bytes, hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess) e = rtr::launch_probe_hits(p, d_o, d_d, n_rays, d_out, ctx->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(out, d_out, out_bytes, hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    (void)hipFree(d);
    if (e != hipSuccess) return fail_hip(ctx, e, "rtr_probe_hits");
    return RTR_OK;
}

int rtr_selftest_sqrt(rtr_context *ctx, uint64_t *mismatches) {
    if (!ctx || !mismatches) return RTR_ERR_INVALID_ARGUMENT;
    if (!ctx->ranks.empty()) ctx = ctx->ranks[0];
    RTR_LOCK(ctx);
    if (ctx->device < 0) return fail(ctx, RTR_ERR_NO_DEVICE, "host-only context: no HIP device bound (there is no CPU fallback)");
    RTR_HIP_TRY(ctx, hipSetDevice(ctx->device));
    RTR_HIP_TRY(ctx, hipMemsetAsync(ctx->d_counters, 0, sizeof(unsigned long long), ctx->stream));
    RTR_HIP_TRY(ctx, rtr::launch_selftest_sqrt(ctx->d_counters, ctx->stream));
    unsigned long long bad = 0;
    RTR_HIP_TRY(ctx, hipMemcpyAsync(&bad, ctx->d_counters, sizeof bad, hipMemcpyDeviceToHost, ctx->stream));
    RTR_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    *mismatches = bad;
    return RTR_OK;
}

int rtr_selftest_div(rtr_context *ctx, float denominator, uint32_t numerator_lo_bits, uint32_t numerator_hi_bits, uint64_t *mismatches) {
    if (!ctx || !mismatches) return RTR_ERR_INVALID_ARGUMENT;
    if (!ctx->ranks.empty()) ctx = ctx->ranks[0];
    RTR_LOCK(ctx);
    if (ctx->device < 0) return fail(ctx, RTR_ERR_NO_DEVICE, "host-only context: no HIP device bound (there is no CPU fallback)");
    if (numerator_lo_bits > numerator_hi_bits) return fail(ctx, RTR_ERR_INVALID_ARGUMENT, "empty numerator range");
    RTR_HIP_TRY(ctx, hipSetDevice(ctx->device));
    RTR_HIP_TRY(ctx, hipMemsetAsync(ctx->d_counters, 0, sizeof(unsigned long long), ctx->stream));
    RTR_HIP_TRY(ctx, rtr::launch_selftest_div(denominator, numerator_lo_bits, numerator_hi_bits, ctx->d_counters, ctx->stream));
    unsigned long long bad = 0;
    RTR_HIP_TRY(ctx, hipMemcpyAsync(&bad, ctx->d_counters, sizeof bad, hipMemcpyDeviceToHost, ctx->stream));
    RTR_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    *mismatches = bad;
    return RTR_OK;
}

int rtr_segment_counts(rtr_context *ctx, uint64_t *out, uint32_t cap) {
    if (!ctx || !out) return RTR_ERR_INVALID_ARGUMENT;
    RTR_LOCK(ctx);
    if (ctx->multi) { // per frame: the GPUs' segments added up
        int n = 0;
        std::vector<uint64_t> tmp(cap ? cap : 1u);
        for (size_t r = 0; r < ctx->ranks.size(); r++) {
            const int k = rtr_segment_counts(ctx->ranks[r], tmp.data(), cap);
            if (k < 0) return fail(ctx, k, ctx->ranks[r]->error);
            if (r == 0) { n = k; std::copy(tmp.begin(), tmp.begin() + k, out); }
            else for (int i = 0; i < std::min(n, k); i++) out[n - 1 - i] += tmp[(size_t)(k - 1 - i)];
        }
        return n;
    }
    if (ctx->device < 0 || ctx->launches == 0) return 0;
    RTR_HIP_TRY(ctx, hipSetDevice(ctx->device));
    RTR_HIP_TRY(ctx, hipDeviceSynchronize());
    if (int rc = check_fault(ctx)) return rc;
    uint32_t n = ctx->launches < (uint32_t)kRing ? ctx->launches : (uint32_t)kRing;
    if (n > cap) n = cap;
    unsigned long long ring[kRing];
    RTR_HIP_TRY(ctx, hipMemcpy(ring, ctx->d_counters, sizeof ring, hipMemcpyDeviceToHost));
    for (uint32_t i = 0; i < n; i++) out[i] = ring[(ctx->launches - n + i) % kRing];
    return (int)n;
}

} // extern "C"
