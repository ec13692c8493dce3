// rtr_lbvh.hip -- device-side BVH builder for mesh subtrees (option mesh_tree = "lbvh").
//
// The reference builds every mesh tree on the CPU by recursive median split with a stable sort per level
// (lib/core/bvh.rs:59-144, O(n log^2 n), rebuilt by Mesh::build_bvh, mesh.rs:119-121).  This is the GPU alternative for
// large or changing meshes: a linear BVH over the SAME per-triangle boxes (mesh.rs:201-213, computed by the host
// flattener in O(n)) -- Morton codes of the box centroids, one radix sort (rocPRIM, rtr_sort.hip), Karras' parallel radix-tree construction
// (one thread per inner node), and one bottom-up pass that unions the boxes, measures the height and writes the 64-byte
// node records of rtr_device.h in place.  A different tree finds the same closest hits (every triangle is still reached
// through boxes that contain it; exact-t ties resolve by triangle handle here, not by the reference's depth-first rank);
// only box-boundary grazes can differ, as with the opt-in SAH trees.
//
// Layout written: inner node j of a mesh -> nodes[node_base + j] (node 0 is the mesh's root); a child that is an inner
// node k -> ref node_base + k with its box; a child that is a triangle h -> the bare leaf ref REF_LEAF | REF_TRI |
// (leaf_base + h), no box (BvhNode::Object has none, bvh.rs:54).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "rtr_device.h"

namespace rtr {

hipError_t sort_keys_u64(void *temp, size_t &temp_bytes, const unsigned long long *keys_in, unsigned long long *keys_out, uint32_t n,
                         uint32_t key_bits, hipStream_t stream); // rtr_sort.hip (rocPRIM radix sort)

namespace {

__device__ __forceinline__ uint32_t expand_bits10(uint32_t v) { // 10 bits -> every third bit
    v = (v * 0x00010001u) & 0xFF0000FFu;
    v = (v * 0x00000101u) & 0x0F00F00Fu;
    v = (v * 0x00000011u) & 0xC30C30C3u;
    v = (v * 0x00000005u) & 0x49249249u;
    return v;
}

// key = 30-bit Morton code of the box centroid inside the mesh's bounds, made unique by the triangle handle
__global__ void k_lbvh_keys(const float *boxes, uint32_t n, float3 bmin, float3 inv_ext, unsigned long long *keys) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float *b = boxes + 6ull * i;
    float cx = ((b[0] + b[3]) * 0.5f - bmin.x) * inv_ext.x, cy = ((b[1] + b[4]) * 0.5f - bmin.y) * inv_ext.y,
          cz = ((b[2] + b[5]) * 0.5f - bmin.z) * inv_ext.z;
    cx = fminf(fmaxf(cx * 1024.0f, 0.0f), 1023.0f);
    cy = fminf(fmaxf(cy * 1024.0f, 0.0f), 1023.0f);
    cz = fminf(fmaxf(cz * 1024.0f, 0.0f), 1023.0f);
    const uint32_t code = (expand_bits10((uint32_t)cx) << 2) | (expand_bits10((uint32_t)cy) << 1) | expand_bits10((uint32_t)cz);
    keys[i] = ((unsigned long long)code << 32) | i;
}

__device__ __forceinline__ int delta(const unsigned long long *keys, int n, int i, int j) { // common prefix length, -1 outside
    if (j < 0 || j >= n) return -1;
    return __clzll((long long)(keys[i] ^ keys[j]));
}

// Karras 2012: inner node i covers a contiguous range of the sorted keys; its split is where the common prefix grows
__global__ void k_lbvh_hierarchy(const unsigned long long *keys, uint32_t n, uint32_t *child, uint32_t *parent) {
    const int i = (int)(blockIdx.x * blockDim.x + threadIdx.x);
    const int inner = (int)n - 1;
    if (i >= inner) return;
    const int d = delta(keys, (int)n, i, i + 1) - delta(keys, (int)n, i, i - 1) >= 0 ? 1 : -1;
    const int dmin = delta(keys, (int)n, i, i - d);
    int lmax = 2;
    while (delta(keys, (int)n, i, i + lmax * d) > dmin) lmax <<= 1;
    int l = 0;
    for (int t = lmax >> 1; t >= 1; t >>= 1)
        if (delta(keys, (int)n, i, i + (l + t) * d) > dmin) l += t;
    const int j = i + l * d;
    const int dnode = delta(keys, (int)n, i, j);
    int s = 0;
    for (int t = (l + 1) >> 1;; t = (t + 1) >> 1) { // ceil(l / 2), ceil(l / 4), ..., 1
        if (delta(keys, (int)n, i, i + (s + t) * d) > dnode) s += t;
        if (t == 1) break;
    }
    const int gamma = i + s * d + (d < 0 ? -1 : 0);
    const int lo = i < j ? i : j, hi = i < j ? j : i;
    // children: bit 31 set = leaf (index into the sorted keys), else inner node
    const uint32_t left = lo == gamma ? (0x80000000u | (uint32_t)gamma) : (uint32_t)gamma;
    const uint32_t right = hi == gamma + 1 ? (0x80000000u | (uint32_t)(gamma + 1)) : (uint32_t)(gamma + 1);
    child[2 * i] = left;
    child[2 * i + 1] = right;
    if (left & 0x80000000u) parent[inner + (int)(left & 0x7FFFFFFFu)] = (uint32_t)i; else parent[left] = (uint32_t)i;
    if (right & 0x80000000u) parent[inner + (int)(right & 0x7FFFFFFFu)] = (uint32_t)i; else parent[right] = (uint32_t)i;
    if (i == 0) parent[0] = 0xFFFFFFFFu;
}

// One thread per leaf walks towards the root; the SECOND thread to reach an inner node finds both children complete,
// unions their boxes, takes the height and writes the node's record.
__global__ void k_lbvh_emit(const unsigned long long *keys, const float *boxes, uint32_t n, const uint32_t *child, const uint32_t *parent,
                            uint32_t *arrived, float *node_box, uint32_t *node_height, DevNode *out, uint32_t node_base,
                            uint32_t leaf_base, uint32_t *height_out) {
    const uint32_t leaf = blockIdx.x * blockDim.x + threadIdx.x;
    if (leaf >= n) return;
    const uint32_t inner = n - 1u;
    uint32_t cur = parent[inner + leaf];
    while (cur != 0xFFFFFFFFu) {
        __threadfence(); // what this thread wrote below (its subtree's box) is visible before it announces itself
        if (atomicAdd(&arrived[cur], 1u) == 0u) return; // the sibling subtree is not finished: its thread carries on
        __threadfence();
        DevNode rec;
        float bx[2][6];
        uint32_t refs[2], h[2];
        for (int c = 0; c < 2; c++) {
            const uint32_t ch = child[2u * cur + c];
            if (ch & 0x80000000u) {
                const uint32_t handle = (uint32_t)(keys[ch & 0x7FFFFFFFu] & 0xFFFFFFFFull);
                const float *b = boxes + 6ull * handle;
                for (int k = 0; k < 6; k++) bx[c][k] = b[k];
                refs[c] = REF_LEAF | REF_TRI | (leaf_base + handle);
                h[c] = 0u;
            } else {
                for (int k = 0; k < 6; k++) bx[c][k] = __hip_atomic_load(&node_box[6ull * ch + k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                refs[c] = node_base + ch;
                h[c] = __hip_atomic_load(&node_height[ch], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
        for (int k = 0; k < 3; k++) {
            rec.lmin[k] = bx[0][k]; rec.lmax[k] = bx[0][3 + k];
            rec.rmin[k] = bx[1][k]; rec.rmax[k] = bx[1][3 + k];
            __hip_atomic_store(&node_box[6ull * cur + k], fminf(bx[0][k], bx[1][k]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(&node_box[6ull * cur + 3 + k], fmaxf(bx[0][3 + k], bx[1][3 + k]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        rec.lref = refs[0]; rec.rref = refs[1]; rec.flags = 0u; rec.pad1 = 0u;
        out[cur] = rec;
        const uint32_t height = 1u + (h[0] > h[1] ? h[0] : h[1]);
        __hip_atomic_store(&node_height[cur], height, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (cur == 0u) atomicMax(height_out, height);
        cur = parent[cur];
    }
}

} // namespace

// Builds the tree of one mesh (n >= 2 triangles).  d_boxes: n x 6 floats (min, max); d_nodes_out: the mesh's n - 1 records
// (= nodes + node_base); *d_height: atomicMax'ed with the tree's height (node records on its longest root-to-leaf path).
hipError_t lbvh_build(const float *d_boxes, uint32_t n, const float bounds_min[3], const float bounds_max[3], DevNode *d_nodes_out,
                      uint32_t node_base, uint32_t leaf_base, uint32_t *d_height, hipStream_t stream) {
    if (n < 2u) return hipErrorInvalidValue;
    const uint32_t inner = n - 1u;
    unsigned long long *keys_in = nullptr, *keys = nullptr;
    uint32_t *child = nullptr, *parent = nullptr, *arrived = nullptr, *node_height = nullptr;
    float *node_box = nullptr;
    void *temp = nullptr;
    size_t temp_bytes = 0;
    hipError_t e = hipSuccess;
    auto done = [&](hipError_t err) {
        (void)hipStreamSynchronize(stream);
        void *bufs[] = {keys_in, keys, child, parent, arrived, node_height, node_box, temp};
        for (void *b : bufs)
            if (b) (void)hipFree(b);
        return err;
    };
    if ((e = hipMalloc((void **)&keys_in, n * sizeof(unsigned long long))) != hipSuccess) return done(e);
    if ((e = hipMalloc((void **)&keys, n * sizeof(unsigned long long))) != hipSuccess) return done(e);
    if ((e = hipMalloc((void **)&child, 2ull * inner * sizeof(uint32_t))) != hipSuccess) return done(e);
    if ((e = hipMalloc((void **)&parent, (size_t)(inner + n) * sizeof(uint32_t))) != hipSuccess) return done(e);
    if ((e = hipMalloc((void **)&arrived, inner * sizeof(uint32_t))) != hipSuccess) return done(e);
    if ((e = hipMalloc((void **)&node_height, inner * sizeof(uint32_t))) != hipSuccess) return done(e);
    if ((e = hipMalloc((void **)&node_box, 6ull * inner * sizeof(float))) != hipSuccess) return done(e);
    if ((e = hipMemsetAsync(arrived, 0, inner * sizeof(uint32_t), stream)) != hipSuccess) return done(e);
    float3 bmin = make_float3(bounds_min[0], bounds_min[1], bounds_min[2]);
    float3 ext = make_float3(bounds_max[0] - bounds_min[0], bounds_max[1] - bounds_min[1], bounds_max[2] - bounds_min[2]);
    float3 inv_ext = make_float3(ext.x > 0.0f ? 1.0f / ext.x : 0.0f, ext.y > 0.0f ? 1.0f / ext.y : 0.0f, ext.z > 0.0f ? 1.0f / ext.z : 0.0f);
    const uint32_t tb = 256u;
    hipLaunchKernelGGL(k_lbvh_keys, dim3((n + tb - 1u) / tb), dim3(tb), 0, stream, d_boxes, n, bmin, inv_ext, keys_in);
    if ((e = hipGetLastError()) != hipSuccess) return done(e);
    if ((e = sort_keys_u64(nullptr, temp_bytes, keys_in, keys, n, 62u, stream)) != hipSuccess) return done(e);
    if ((e = hipMalloc(&temp, temp_bytes ? temp_bytes : 16)) != hipSuccess) return done(e);
    if ((e = sort_keys_u64(temp, temp_bytes, keys_in, keys, n, 62u, stream)) != hipSuccess) return done(e);
    hipLaunchKernelGGL(k_lbvh_hierarchy, dim3((inner + tb - 1u) / tb), dim3(tb), 0, stream, keys, n, child, parent);
    if ((e = hipGetLastError()) != hipSuccess) return done(e);
    hipLaunchKernelGGL(k_lbvh_emit, dim3((n + tb - 1u) / tb), dim3(tb), 0, stream, keys, d_boxes, n, child, parent, arrived, node_box,
                       node_height, d_nodes_out, node_base, leaf_base, d_height);
    if ((e = hipGetLastError()) != hipSuccess) return done(e);
    return done(hipSuccess);
}

} // namespace rtr
