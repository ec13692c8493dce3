// rtr_scene.cpp -- scene lowering, the reference's BVH builder and the flattener (host, no HIP).
//
// The tree SHAPE is part of the contract: exact-t ties resolve by depth-first leaf order (bvh.rs:181-186)
// and the set of boxes decides which triangles are ever tested, so the builder reproduces
// BvhNode::from_list (lib/core/bvh.rs:59-144) decision for decision:
//   bounds   = reduce(surround) over the members' bounds, seeded by the first member (bvh.rs:66-70)
//   axis     = x iff sx>sy && sx>sz; else y iff sy>sx && sy>sz; else z, s = spread of the member-bounds
//              centroids (bvh.rs:73-87)
//   order    = stable ascending sort on centroid[axis] (bvh.rs:90-123); NaN -> the reference panics
//   split    = first floor(n/2) members left, rest right (bvh.rs:126-135); n==1 -> leaf, n==0 -> None
// Built with -ffp-contract=off: bounds, centroids, edges and face normals must round as the reference's f32.
#include "rtr_scene.h"

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <numeric>

namespace rtr {
namespace {

struct Box {
    float mn[3], mx[3];
};

// AABB::surround (lib/utils/aabb.rs:41-54): f32::min / f32::max per component
inline Box surround(const Box &a, const Box &b) {
    Box r;
    for (int k = 0; k < 3; k++) {
        r.mn[k] = std::fmin(a.mn[k], b.mn[k]);
        r.mx[k] = std::fmax(a.mx[k], b.mx[k]);
    }
    return r;
}
inline Box point_box(const float *p) {
    Box r;
    for (int k = 0; k < 3; k++) r.mn[k] = r.mx[k] = p[k];
    return r;
}
// AABB::centroid (aabb.rs:28-30)
inline float centroid(const Box &b, int k) { return (b.mn[k] + b.mx[k]) / 2.0f; }

// AABB::epsilon_expand (aabb.rs:57-78)
inline Box epsilon_expand(const Box &b, float eps) {
    Box r = b;
    for (int k = 0; k < 3; k++) {
        float dim = b.mx[k] - b.mn[k];
        float c = centroid(b, k);
        if (dim < eps) {
            r.mn[k] = c - eps;
            r.mx[k] = c + eps;
        }
    }
    return r;
}

enum : uint8_t { T_NONE = 0, T_LEAF = 1, T_SPLIT = 2 };
struct TreeNode {
    uint8_t kind = T_NONE;
    uint32_t handle = 0;
    Box box{};
    int32_t left = -1, right = -1;
};

// One BoundsCollection (bvh.rs:18-28) reduced to what the builder reads: bounds(handle).
struct Tree {
    std::vector<TreeNode> nodes;
    int32_t root = -1;
};

class Builder {
  public:
    Builder(const std::vector<Box> &bounds, Tree &tree) : bounds_(bounds), tree_(tree) {}
    bool nan_seen = false;

    int32_t build(std::vector<uint32_t> &h, size_t b, size_t e) {
        int32_t id = (int32_t)tree_.nodes.size();
        tree_.nodes.emplace_back();
        size_t n = e - b;
        if (n == 0) return id;
        if (n == 1) {
            tree_.nodes[id].kind = T_LEAF;
            tree_.nodes[id].handle = h[b];
            return id;
        }
        Box all = bounds_[h[b]];
        for (size_t i = b + 1; i < e; i++) all = surround(all, bounds_[h[i]]);
        float cmin[3], cmax[3];
        for (int k = 0; k < 3; k++) cmin[k] = cmax[k] = centroid(bounds_[h[b]], k);
        for (size_t i = b + 1; i < e; i++)
            for (int k = 0; k < 3; k++) {
                float c = centroid(bounds_[h[i]], k);
                cmin[k] = std::fmin(cmin[k], c);
                cmax[k] = std::fmax(cmax[k], c);
            }
        float sx = cmax[0] - cmin[0], sy = cmax[1] - cmin[1], sz = cmax[2] - cmin[2];
        int axis = (sx > sy && sx > sz) ? 0 : ((sy > sx && sy > sz) ? 1 : 2);
        for (size_t i = b; i < e; i++) {
            float c = centroid(bounds_[h[i]], axis);
            if (c != c) nan_seen = true;
        }
        if (!nan_seen)
            std::stable_sort(h.begin() + (std::ptrdiff_t)b, h.begin() + (std::ptrdiff_t)e, [&](uint32_t p, uint32_t q) {
                return centroid(bounds_[p], axis) < centroid(bounds_[q], axis);
            });
        size_t mid = b + n / 2;
        int32_t l = build(h, b, mid);
        int32_t r = build(h, mid, e);
        TreeNode &nd = tree_.nodes[id];
        nd.kind = T_SPLIT;
        nd.box = all;
        nd.left = l;
        nd.right = r;
        return id;
    }

  private:
    const std::vector<Box> &bounds_;
    Tree &tree_;
};

struct MeshBuild {
    Tree tree;                     // the reference's tree (mesh.rs:119-121): parity walk, and the leaf ranks
    Tree fast;                     // optional: binned-SAH tree over the same triangle boxes (BuildOptions)
    bool use_fast = false;
    std::vector<uint32_t> rank;    // per triangle handle: its rank among the mesh's leaves in the REFERENCE tree's
                                   // depth-first order -- exact-t ties resolve by it whichever tree is walked
    std::vector<DevLeafHot> hot;   // per triangle handle
    std::vector<DevLeafCold> cold; // per triangle handle
    std::vector<Box> tri_boxes;    // per triangle handle (mesh.rs:201-213), kept for the device builder
    bool lbvh = false;             // the subtree is built on the device (BuildOptions::lbvh_mesh_trees)
    Box bounds;                    // Mesh.bounds, mesh.rs:99-108
};

inline float half_area(const Box &b) {
    float dx = b.mx[0] - b.mn[0], dy = b.mx[1] - b.mn[1], dz = b.mx[2] - b.mn[2];
    return dx * dy + dy * dz + dz * dx;
}

// Binned surface-area-heuristic builder over the SAME leaf boxes (one triangle per leaf, so the node and leaf
// formats are unchanged).  A different topology changes only how much of the tree a ray visits: the closest hit is
// the same, exact-t ties are broken by reference rank, and the root box (the union of all leaf boxes) is identical.
// Beyond `depth_limit` it splits by median so that the traversal stack (tree height) stays bounded.
class SahBuilder {
  public:
    SahBuilder(const std::vector<Box> &bounds, Tree &tree, uint32_t depth_limit) : bounds_(bounds), tree_(tree), limit_(depth_limit) {}
    int32_t build(std::vector<uint32_t> &h, size_t b, size_t e, uint32_t depth) {
        int32_t id = (int32_t)tree_.nodes.size();
        tree_.nodes.emplace_back();
        size_t n = e - b;
        if (n == 0) return id;
        if (n == 1) { tree_.nodes[id].kind = T_LEAF; tree_.nodes[id].handle = h[b]; return id; }
        Box all = bounds_[h[b]];
        float cmin[3], cmax[3];
        for (int k = 0; k < 3; k++) cmin[k] = cmax[k] = centroid(bounds_[h[b]], k);
        for (size_t i = b + 1; i < e; i++) {
            all = surround(all, bounds_[h[i]]);
            for (int k = 0; k < 3; k++) { float c = centroid(bounds_[h[i]], k); cmin[k] = std::fmin(cmin[k], c); cmax[k] = std::fmax(cmax[k], c); }
        }
        size_t mid = b + n / 2;
        int best_axis = -1;
        float best_pos = 0.0f;
        if (depth < limit_) {
            constexpr int NB = 16;
            float best_cost = INFINITY;
            for (int ax = 0; ax < 3; ax++) {
                float lo = cmin[ax], hi = cmax[ax];
                if (!(hi > lo)) continue;
                Box bb[NB]; uint32_t cnt[NB] = {0};
                for (size_t i = b; i < e; i++) {
                    int bi = std::min(NB - 1, (int)((centroid(bounds_[h[i]], ax) - lo) / (hi - lo) * NB));
                    bb[bi] = cnt[bi] ? surround(bb[bi], bounds_[h[i]]) : bounds_[h[i]];
                    cnt[bi]++;
                }
                for (int sp = 1; sp < NB; sp++) {
                    Box L{}, R{}; uint32_t nl = 0, nr = 0;
                    for (int i = 0; i < sp; i++) if (cnt[i]) { L = nl ? surround(L, bb[i]) : bb[i]; nl += cnt[i]; }
                    for (int i = sp; i < NB; i++) if (cnt[i]) { R = nr ? surround(R, bb[i]) : bb[i]; nr += cnt[i]; }
                    if (!nl || !nr) continue;
                    float cost = half_area(L) * (float)nl + half_area(R) * (float)nr;
                    if (cost < best_cost) { best_cost = cost; best_axis = ax; best_pos = lo + (hi - lo) * (float)sp / (float)NB; }
                }
            }
        }
        if (best_axis >= 0) {
            auto it = std::stable_partition(h.begin() + (std::ptrdiff_t)b, h.begin() + (std::ptrdiff_t)e,
                                            [&](uint32_t t) { return centroid(bounds_[t], best_axis) < best_pos; });
            size_t m = (size_t)(it - h.begin());
            if (m > b && m < e) mid = m;
        } else { // median on the widest centroid axis (also the fallback below the depth limit)
            int ax = 0;
            for (int k = 1; k < 3; k++) if (cmax[k] - cmin[k] > cmax[ax] - cmin[ax]) ax = k;
            std::stable_sort(h.begin() + (std::ptrdiff_t)b, h.begin() + (std::ptrdiff_t)e,
                             [&](uint32_t p, uint32_t q) { return centroid(bounds_[p], ax) < centroid(bounds_[q], ax); });
        }
        int32_t l = build(h, b, mid, depth + 1);
        int32_t r = build(h, mid, e, depth + 1);
        TreeNode &nd = tree_.nodes[id];
        nd.kind = T_SPLIT; nd.box = all; nd.left = l; nd.right = r;
        return id;
    }
  private:
    const std::vector<Box> &bounds_;
    Tree &tree_;
    uint32_t limit_;
};

void leaf_ranks(const Tree &t, int32_t ni, std::vector<uint32_t> &rank, uint32_t &next) {
    const TreeNode &n = t.nodes[ni];
    if (n.kind == T_LEAF) rank[n.handle] = next++;
    else if (n.kind == T_SPLIT) { leaf_ranks(t, n.left, rank, next); leaf_ranks(t, n.right, rank, next); }
}
uint32_t tree_height(const Tree &t, int32_t ni) {
    const TreeNode &n = t.nodes[ni];
    return n.kind == T_SPLIT ? 1u + std::max(tree_height(t, n.left), tree_height(t, n.right)) : 0u;
}

inline void sub3(const float *a, const float *b, float *o) {
    o[0] = a[0] - b[0];
    o[1] = a[1] - b[1];
    o[2] = a[2] - b[2];
}
inline void cross3(const float *a, const float *b, float *o) { // cgmath Vector3::cross
    o[0] = (a[1] * b[2]) - (a[2] * b[1]);
    o[1] = (a[2] * b[0]) - (a[0] * b[2]);
    o[2] = (a[0] * b[1]) - (a[1] * b[0]);
}

struct Flattener {
    const rtr_scene_desc &desc;
    const std::vector<MeshBuild> &meshes;
    const Tree &scene_tree;
    FlatScene &out;
    std::vector<uint32_t> &obj_ref;    // per object: ref of its leaf / subtree root
    std::vector<Box> &obj_box;         // per object: box of that root when it is a Split
    std::vector<uint32_t> &obj_height; // per object: Split nodes on its longest root-to-leaf path
    const std::vector<uint32_t> &obj_leaf_base; // per object: first leaf slot = its rank block in the REFERENCE depth-first order
    const std::vector<uint32_t> &vol_leaf_base; // per volume: first leaf slot of its boundary mesh's triangles (mesh boundaries)
    const std::vector<bool> &vol_nested;        // per volume: its boundary is a Volume (lowered onto that Volume's boundary)
    const std::vector<float> &vol_inner_density; // ... whose -1/density this is
    bool leaf_pairs = true;
    std::vector<uint32_t> obj_pushes; // per object: node records on its longest root-to-leaf path

    // Box of a flattened ref's own Split node, needed by its parent.
    struct Emitted {
        uint32_t ref;
        Box box;
        uint32_t height;
        bool has_volume = false; // the subtree holds a Volume leaf
        uint32_t pushes = 0;     // node records on the longest path: what the walk can push (a leaf pair has no record)
    };

    // Volume (lib/volume.rs:19-32).  Sphere boundary: hot = centre, radius, radius^2, -1/density, kind 0.
    // Mesh boundary: hot = the mesh's root box (f[0..2] min, f[9..11] max), -1/density, kind 1, the root ref of the mesh's
    // subtree (always the reference's tree: Volume::hit walks it with ranges that start below zero, where only the
    // reference's unculled visit order is defined); the triangles' leaf records sit behind the scene tree's leaves.
    Emitted emit_volume(uint32_t object, uint32_t vi, const rtr_volume &v) {
        uint32_t L = obj_leaf_base[object];
        DevLeafHot hot{};
        DevLeafCold cold{};
        uint32_t pushes = 0;
        if (v.boundary_kind == RTR_OBJ_MESH) {
            const MeshBuild &mb = meshes[v.boundary_mesh];
            Emitted e = emit_mesh_node(object, mb, mb.tree, mb.tree.root, vol_leaf_base[vi]);
            hot.f[0] = e.box.mn[0]; hot.f[1] = e.box.mn[1]; hot.f[2] = e.box.mn[2];
            hot.f[9] = e.box.mx[0]; hot.f[10] = e.box.mx[1]; hot.f[11] = e.box.mx[2];
            const uint32_t kind = 1u;
            std::memcpy(&hot.f[6], &kind, 4);
            std::memcpy(&hot.f[7], &e.ref, 4);
            cold.material = desc.meshes[v.boundary_mesh].material;
            pushes = e.pushes + 1u; // the nested walk continues on the lane's stack, above the outer walk's entries
            out.has_mesh_volumes = true;
        } else {
            const rtr_sphere &s = v.boundary;
            hot.f[0] = s.center[0]; hot.f[1] = s.center[1]; hot.f[2] = s.center[2]; hot.f[3] = s.radius;
            hot.f[4] = s.radius * s.radius;
            cold.material = s.material;
        }
        hot.f[5] = v.neg_inv_density;
        if (vol_nested[vi]) { // a Volume inside a Volume: bit 8 of the kind word, the inner Volume's -1/density in the spare word
            uint32_t kind;
            std::memcpy(&kind, &hot.f[6], 4);
            kind |= 0x100u;
            std::memcpy(&hot.f[6], &kind, 4);
            hot.f[8] = vol_inner_density[vi];
            out.has_nested_volumes = true;
        }
        cold.object = object; cold.handle = 0; // Object::material = boundary.material(), volume.rs:98-100
        out.leaf_hot[L] = hot;
        out.leaf_cold[L] = cold;
        out.has_volumes = true;
        return {REF_LEAF | REF_VOL | L, Box{}, 0, true, pushes};
    }

    Emitted emit_sphere(uint32_t object, const rtr_sphere &s) {
        uint32_t L = obj_leaf_base[object];
        DevLeafHot hot{};
        // centre and r^2 -- all Sphere::hit reads -- in the first 16 bytes: a leaf test is ONE load instruction (the leaf block of
        // a trip runs with few lanes, where a gather costs per instruction: DESIGN.md section 5.2); the radius is read once, by the winner
        hot.f[0] = s.center[0]; hot.f[1] = s.center[1]; hot.f[2] = s.center[2];
        hot.f[3] = s.radius * s.radius; // objects.rs:27 `self.radius * self.radius`
        hot.f[4] = s.radius;
        DevLeafCold cold{};
        cold.material = s.material; cold.object = object; cold.handle = 0;
        out.leaf_hot[L] = hot;
        out.leaf_cold[L] = cold;
        return {REF_LEAF | L, Box{}, 0};
    }

    // Leaf records of a mesh occupy one block, ordered by REFERENCE depth-first rank whichever tree is flattened.
    Emitted emit_mesh_node(uint32_t object, const MeshBuild &mb, const Tree &t, int32_t ni, uint32_t leaf_base) {
        const TreeNode &n = t.nodes[ni];
        if (n.kind == T_NONE) return {REF_NONE, Box{}, 0};
        if (n.kind == T_LEAF) {
            uint32_t L = leaf_base + mb.rank[n.handle];
            out.leaf_hot[L] = mb.hot[n.handle];
            DevLeafCold c = mb.cold[n.handle];
            c.object = object;
            out.leaf_cold[L] = c;
            return {REF_LEAF | REF_TRI | L, Box{}, 0};
        }
        out.split_nodes++;
        if (leaf_pairs && t.nodes[n.left].kind == T_LEAF && t.nodes[n.right].kind == T_LEAF &&
            mb.rank[t.nodes[n.right].handle] == mb.rank[t.nodes[n.left].handle] + 1u) {
            // Split(triangle L, triangle L+1): no node record, the parent gates the pair with this Split's box
            Emitted l = emit_mesh_node(object, mb, t, n.left, leaf_base);
            emit_mesh_node(object, mb, t, n.right, leaf_base);
            return {l.ref | REF_PAIR, n.box, 1};
        }
        uint32_t idx = (uint32_t)out.nodes.size();
        out.nodes.emplace_back();
        Emitted l = emit_mesh_node(object, mb, t, n.left, leaf_base);
        Emitted r = emit_mesh_node(object, mb, t, n.right, leaf_base);
        fill_node(idx, l, r);
        return {idx, n.box, 1 + std::max(l.height, r.height), false, 1 + std::max(l.pushes, r.pushes)};
    }

    // A mesh whose subtree the device builds: leaf records in handle order, n_tris - 1 node records reserved, the job noted.
    Emitted emit_lbvh_mesh(uint32_t object, const MeshBuild &mb, uint32_t leaf_base) {
        const uint32_t n = (uint32_t)mb.hot.size();
        for (uint32_t h = 0; h < n; h++) {
            out.leaf_hot[leaf_base + h] = mb.hot[h];
            DevLeafCold c = mb.cold[h];
            c.object = object;
            out.leaf_cold[leaf_base + h] = c;
        }
        LbvhJob job{};
        job.node_base = (uint32_t)out.nodes.size();
        job.n_tris = n;
        job.leaf_base = leaf_base;
        job.box_offset = out.lbvh_boxes.size() / 6u;
        Box all = mb.tri_boxes[0];
        for (const Box &bx : mb.tri_boxes) {
            all = surround(all, bx);
            for (int k = 0; k < 3; k++) out.lbvh_boxes.push_back(bx.mn[k]);
            for (int k = 0; k < 3; k++) out.lbvh_boxes.push_back(bx.mx[k]);
        }
        for (int k = 0; k < 3; k++) { job.bounds_min[k] = all.mn[k]; job.bounds_max[k] = all.mx[k]; }
        out.nodes.resize(out.nodes.size() + (n - 1u)); // written by the device builder
        out.split_nodes += n - 1u;
        out.lbvh_jobs.push_back(job);
        return {job.node_base, all, 0, false, 0}; // height / pushes: measured on the device, added by rtr_upload_scene
    }

    Emitted emit_object(uint32_t object) {
        const rtr_object &o = desc.objects[object];
        Emitted e;
        if (o.kind == RTR_OBJ_SPHERE) {
            e = emit_sphere(object, desc.spheres[o.index]);
        } else if (o.kind == RTR_OBJ_VOLUME) {
            e = emit_volume(object, o.index, desc.volumes[o.index]);
        } else {
            const MeshBuild &mb = meshes[o.index];
            const uint32_t leaf_base = obj_leaf_base[object];
            if (mb.lbvh) {
                e = emit_lbvh_mesh(object, mb, leaf_base);
            } else {
                const Tree &t = mb.use_fast ? mb.fast : mb.tree;
                e = emit_mesh_node(object, mb, t, t.root, leaf_base); // Mesh::hit = bvh_root.hit, mesh.rs:124-127
            }
        }
        obj_ref[object] = e.ref;
        obj_box[object] = e.box;
        obj_height[object] = e.height;
        if (obj_pushes.size() < desc.n_objects) obj_pushes.resize(desc.n_objects, 0u);
        obj_pushes[object] = e.pushes;
        return e;
    }

    bool is_sphere_leaf(int32_t ni) const {
        const TreeNode &n = scene_tree.nodes[ni];
        return n.kind == T_LEAF && desc.objects[n.handle].kind == RTR_OBJ_SPHERE;
    }

    Emitted emit_scene_node(int32_t ni) {
        const TreeNode &n = scene_tree.nodes[ni];
        if (n.kind == T_NONE) return {REF_NONE, Box{}, 0};
        if (n.kind == T_LEAF) return emit_object(n.handle);
        out.split_nodes++;
        if (leaf_pairs && is_sphere_leaf(n.left) && is_sphere_leaf(n.right) &&
            obj_leaf_base[scene_tree.nodes[n.right].handle] == obj_leaf_base[scene_tree.nodes[n.left].handle] + 1u) {
            Emitted l = emit_scene_node(n.left); // Split(sphere L, sphere L+1)
            emit_scene_node(n.right);
            return {l.ref | REF_PAIR, n.box, 1};
        }
        uint32_t idx = (uint32_t)out.nodes.size();
        out.nodes.emplace_back();
        Emitted l = emit_scene_node(n.left);
        Emitted r = emit_scene_node(n.right);
        fill_node(idx, l, r);
        return {idx, n.box, 1 + std::max(l.height, r.height), l.has_volume || r.has_volume, 1 + std::max(l.pushes, r.pushes)};
    }

    void fill_node(uint32_t idx, const Emitted &l, const Emitted &r) {
        DevNode &d = out.nodes[idx];
        std::memset(&d, 0, sizeof d);
        for (int k = 0; k < 3; k++) {
            d.lmin[k] = l.box.mn[k]; d.lmax[k] = l.box.mx[k];
            d.rmin[k] = r.box.mn[k]; d.rmax[k] = r.box.mx[k];
        }
        d.lref = l.ref;
        d.rref = r.ref;
        d.flags = (l.has_volume ? NODE_LVOL : 0u) | (r.has_volume ? NODE_RVOL : 0u);
    }
};

} // namespace

int build_flat_scene(const rtr_scene_desc &desc_in, FlatScene &out, std::string &err, const BuildOptions &opts) {
    out = FlatScene{};
    if ((desc_in.n_objects && !desc_in.objects) || (desc_in.n_spheres && !desc_in.spheres) || (desc_in.n_meshes && !desc_in.meshes) ||
        (desc_in.n_materials && !desc_in.materials) || (desc_in.n_volumes && !desc_in.volumes)) {
        err = "null array with non-zero count in rtr_scene_desc";
        return RTR_ERR_INVALID_ARGUMENT;
    }
    // ---- Volumes whose boundary is a Volume (lib/volume.rs:20 `boundary: Box<dyn Object>`) ----
    // Volume::hit asks its boundary twice (volume.rs:35-36); when the boundary is a Volume those two calls are Volume::hit
    // again, on the inner boundary.  Bounds and material of the outer Volume are the innermost boundary's (volume.rs:90-100),
    // so the pair lowers to ONE leaf: the inner Volume's sphere or mesh, the OUTER density in the record's usual word and the
    // inner density beside it (`inner_density`); the kernels evaluate the nested calls.  One level of nesting.
    std::vector<rtr_volume> vols(desc_in.volumes, desc_in.volumes + desc_in.n_volumes);
    std::vector<float> inner_density(desc_in.n_volumes, 0.0f);
    std::vector<bool> nested(desc_in.n_volumes, false), volume_is_object(desc_in.n_volumes, false);
    for (uint32_t i = 0; i < desc_in.n_objects; i++)
        if (desc_in.objects[i].kind == RTR_OBJ_VOLUME && desc_in.objects[i].index < desc_in.n_volumes) volume_is_object[desc_in.objects[i].index] = true;
    for (uint32_t vi = 0; vi < desc_in.n_volumes; vi++) {
        if (desc_in.volumes[vi].boundary_kind != RTR_OBJ_VOLUME) continue;
        const uint32_t inner = desc_in.volumes[vi].boundary_mesh; // (the index field names the boundary Volume here)
        if (inner >= desc_in.n_volumes || inner == vi) { err = "volume boundary volume out of bounds"; return RTR_ERR_BAD_INDEX; }
        if (desc_in.volumes[inner].boundary_kind == RTR_OBJ_VOLUME) {
            err = "Volumes nested more than one level deep are outside the GPU path";
            return RTR_ERR_UNSUPPORTED;
        }
        vols[vi] = desc_in.volumes[inner];                       // boundary, bounds and material: the innermost boundary's
        vols[vi].neg_inv_density = desc_in.volumes[vi].neg_inv_density;
        inner_density[vi] = desc_in.volumes[inner].neg_inv_density;
        nested[vi] = true;
    }
    rtr_scene_desc desc = desc_in;
    desc.volumes = vols.data();
    // ---- materials (scene.rs:100-110) ----
    out.materials.resize(desc.n_materials);
    for (uint32_t i = 0; i < desc.n_materials; i++) {
        const rtr_material &m = desc.materials[i];
        if (m.kind > RTR_MAT_ISOTROPIC) { err = "unknown material kind"; return RTR_ERR_INVALID_ARGUMENT; }
        DevMaterial d{};
        d.rgb[0] = m.rgb[0]; d.rgb[1] = m.rgb[1]; d.rgb[2] = m.rgb[2];
        d.param = m.param; d.kind = m.kind; d.first = m.first; d.second = m.second;
        if (m.kind == RTR_MAT_MIX) {
            // nested boxes flattened to indices; children after the parent so the device walk terminates
            if (m.first >= desc.n_materials || m.second >= desc.n_materials) { err = "mix child out of bounds"; return RTR_ERR_BAD_INDEX; }
            if (m.first <= i || m.second <= i) { err = "mix children must follow their parent in the materials array"; return RTR_ERR_INVALID_ARGUMENT; }
        }
        if (m.kind == RTR_MAT_DIELECTRIC) {
            // Dielectric::scatter divides twice per hit by constants of the material (materials.rs:113-118 `1.0 / ir`,
            // materials.rs:104-105 `((1 - ratio) / (1 + ratio))^2` for ratio = 1/ir or ir).  The quotients are computed
            // here once, with the same f32 operations (IEEE division is correctly rounded on the host and on the
            // device alike), into the record's otherwise unused words: first = 1/ir, second = r0^2 for the front face
            // (ratio = 1/ir), pad = r0^2 for the back face (ratio = ir).
            const float ir = m.param;
            const float inv_ir = 1.0f / ir;
            float r0_front = (1.0f - inv_ir) / (1.0f + inv_ir);
            r0_front = r0_front * r0_front;
            float r0_back = (1.0f - ir) / (1.0f + ir);
            r0_back = r0_back * r0_back;
            std::memcpy(&d.first, &inv_ir, 4);
            std::memcpy(&d.second, &r0_front, 4);
            std::memcpy(&d.pad, &r0_back, 4);
        }
        out.materials[i] = d;
    }
    // ---- background (backgrounds.rs) ----
    const rtr_background &bg = desc.background;
    out.bg_kind = bg.kind;
    for (int k = 0; k < 3; k++) { out.bg_a[k] = bg.a[k]; out.bg_b[k] = bg.b[k]; }
    if (bg.kind == RTR_BG_SKYMAP) {
        if (!bg.image || bg.width == 0 || bg.height == 0) { err = "empty sky image (% width panics, backgrounds.rs:88)"; return RTR_ERR_INVALID_ARGUMENT; }
        size_t n = (size_t)bg.width * bg.height;
        out.sky.resize(n * 4);
        for (size_t i = 0; i < n; i++) {
            out.sky[4 * i + 0] = bg.image[3 * i + 0];
            out.sky[4 * i + 1] = bg.image[3 * i + 1];
            out.sky[4 * i + 2] = bg.image[3 * i + 2];
            out.sky[4 * i + 3] = 0.0f;
        }
        out.sky_w = bg.width; out.sky_h = bg.height;
    } else if (bg.kind > RTR_BG_SKYMAP) {
        err = "unknown background kind";
        return RTR_ERR_INVALID_ARGUMENT;
    }

    // ---- meshes: Mesh::from_buffers + build_bvh (mesh.rs:78-121) ----
    std::vector<MeshBuild> meshes(desc.n_meshes);
    std::vector<bool> bounds_a_volume(desc.n_meshes, false); // such meshes always get the reference's tree (nested walk)
    for (uint32_t vi = 0; vi < desc.n_volumes; vi++)
        if (volume_is_object[vi] && desc.volumes[vi].boundary_kind == RTR_OBJ_MESH && desc.volumes[vi].boundary_mesh < desc.n_meshes)
            bounds_a_volume[desc.volumes[vi].boundary_mesh] = true;
    for (uint32_t mi = 0; mi < desc.n_meshes; mi++) {
        const rtr_mesh &m = desc.meshes[mi];
        MeshBuild &mb = meshes[mi];
        if ((m.n_verts && (!m.positions || !m.normals)) || (m.n_tris && !m.indices)) { err = "null mesh buffer"; return RTR_ERR_INVALID_ARGUMENT; }
        if (m.material >= desc.n_materials) { err = "mesh material out of bounds"; return RTR_ERR_BAD_INDEX; }
        if (m.n_tris > REF_INDEX) { err = "too many triangles"; return RTR_ERR_INVALID_ARGUMENT; }
        for (uint64_t i = 0; i < 3ull * m.n_tris; i++)
            if (m.indices[i] >= m.n_verts) { err = "vertex index out of bounds (mesh.rs:84-88 panics)"; return RTR_ERR_BAD_INDEX; }
        mb.hot.resize(m.n_tris);
        mb.cold.resize(m.n_tris);
        std::vector<Box> tb(m.n_tris);
        for (uint32_t t = 0; t < m.n_tris; t++) {
            const uint32_t *ix = m.indices + 3ull * t;
            const float *p0 = m.positions + 3ull * ix[0], *p1 = m.positions + 3ull * ix[1], *p2 = m.positions + 3ull * ix[2];
            DevLeafHot &h = mb.hot[t];
            std::memset(&h, 0, sizeof h);
            float e1[3], e2[3];
            sub3(p1, p0, e1); // mesh.rs:160
            sub3(p2, p0, e2); // mesh.rs:161
            h.f[0] = p0[0]; h.f[1] = p0[1]; h.f[2] = p0[2];
            h.f[3] = e1[0]; h.f[4] = e1[1]; h.f[5] = e1[2];
            h.f[6] = e2[0]; h.f[7] = e2[1]; h.f[8] = e2[2];
            // face normal, mesh.rs:86-90: ((v2-v1) x (v0-v1)).normalize()
            float a[3], b[3], c[3];
            sub3(p0, p1, a);
            sub3(p2, p1, b);
            cross3(b, a, c);
            float inv = 1.0f / std::sqrt((c[0] * c[0] + c[1] * c[1]) + c[2] * c[2]);
            DevLeafCold &cd = mb.cold[t];
            std::memset(&cd, 0, sizeof cd);
            cd.fn[0] = c[0] * inv; cd.fn[1] = c[1] * inv; cd.fn[2] = c[2] * inv;
            const float *n0 = m.normals + 3ull * ix[0], *n1 = m.normals + 3ull * ix[1], *n2 = m.normals + 3ull * ix[2];
            for (int k = 0; k < 3; k++) { cd.n0[k] = n0[k]; cd.n1[k] = n1[k]; cd.n2[k] = n2[k]; }
            cd.material = m.material;
            cd.handle = t;
            // BoundsCollection::bounds for Mesh, mesh.rs:201-213
            tb[t] = epsilon_expand(surround(point_box(p0), surround(point_box(p1), point_box(p2))), 0.01f);
        }
        // Mesh.bounds seeded with the zero box (mesh.rs:99-108, aabb.rs:140-147)
        Box mbounds{};
        for (uint32_t v = 0; v < m.n_verts; v++) mbounds = surround(mbounds, point_box(m.positions + 3ull * v));
        mb.bounds = mbounds;
        std::vector<uint32_t> handles(m.n_tris);
        std::iota(handles.begin(), handles.end(), 0u);
        if (opts.lbvh_mesh_trees && m.n_tris >= 2 && !bounds_a_volume[mi]) {
            // the device builds this subtree (rtr_lbvh.hip): no host build, leaf records in handle order
            for (const Box &bx : tb)
                for (int k = 0; k < 3; k++)
                    if (bx.mn[k] != bx.mn[k] || bx.mx[k] != bx.mx[k]) { err = "NaN triangle box"; return RTR_ERR_NAN_CENTROID; }
            mb.lbvh = true;
            mb.tri_boxes = tb;
            mb.rank.resize(m.n_tris);
            std::iota(mb.rank.begin(), mb.rank.end(), 0u);
            continue;
        }
        Builder b(tb, mb.tree);
        mb.tree.root = b.build(handles, 0, handles.size());
        if (b.nan_seen) { err = "NaN triangle centroid: partial_cmp().unwrap() panics (bvh.rs:99,109,119)"; return RTR_ERR_NAN_CENTROID; }
        mb.rank.assign(m.n_tris, 0u);
        uint32_t next = 0;
        leaf_ranks(mb.tree, mb.tree.root, mb.rank, next);
        if (opts.sah_mesh_trees && m.n_tris > 2) {
            std::iota(handles.begin(), handles.end(), 0u);
            // SAH splits down to the balanced tree's height, median splits below: deeper SAH levels cost more LDS
            // stack (occupancy) than they save (measured on house.obj / monkey.obj: limit +2..+30 and -2..-8 all slower)
            SahBuilder sb(tb, mb.fast, tree_height(mb.tree, mb.tree.root));
            mb.fast.root = sb.build(handles, 0, handles.size(), 0);
            mb.use_fast = true;
        }
    }

    // ---- scene-level collection (scene.rs:117-139) ----
    std::vector<Box> ob(desc.n_objects);
    for (uint32_t i = 0; i < desc.n_objects; i++) {
        const rtr_object &o = desc.objects[i];
        if (o.kind == RTR_OBJ_SPHERE) {
            if (o.index >= desc.n_spheres) { err = "sphere index out of bounds"; return RTR_ERR_BAD_INDEX; }
            const rtr_sphere &s = desc.spheres[o.index];
            if (s.material >= desc.n_materials) { err = "sphere material out of bounds (scene.rs:106-110 panics)"; return RTR_ERR_BAD_INDEX; }
            for (int k = 0; k < 3; k++) { // Sphere::bounds, objects.rs:53-60
                ob[i].mn[k] = s.center[k] - s.radius;
                ob[i].mx[k] = s.center[k] + s.radius;
            }
        } else if (o.kind == RTR_OBJ_VOLUME) {
            if (o.index >= desc.n_volumes) { err = "volume index out of bounds"; return RTR_ERR_BAD_INDEX; }
            const rtr_volume &v = desc.volumes[o.index];
            if (v.boundary_kind == RTR_OBJ_MESH) {
                if (v.boundary_mesh >= desc.n_meshes) { err = "volume boundary mesh out of bounds"; return RTR_ERR_BAD_INDEX; }
                ob[i] = meshes[v.boundary_mesh].bounds; // Bounded for Volume = boundary.bounds(), volume.rs:90-94
            } else if (v.boundary_kind == RTR_OBJ_SPHERE) {
                const rtr_sphere &s = v.boundary;
                if (s.material >= desc.n_materials) { err = "volume material out of bounds"; return RTR_ERR_BAD_INDEX; }
                for (int k = 0; k < 3; k++) {
                    ob[i].mn[k] = s.center[k] - s.radius;
                    ob[i].mx[k] = s.center[k] + s.radius;
                }
            } else {
                err = "a Volume's boundary is a sphere or a mesh";
                return RTR_ERR_UNSUPPORTED;
            }
            out.spheres_only = false;
        } else if (o.kind == RTR_OBJ_MESH) {
            if (o.index >= desc.n_meshes) { err = "mesh index out of bounds"; return RTR_ERR_BAD_INDEX; }
            ob[i] = meshes[o.index].bounds; // Bounded for Mesh, mesh.rs:234-238
            out.spheres_only = false;
        } else {
            err = "unknown object kind";
            return RTR_ERR_INVALID_ARGUMENT;
        }
    }
    Tree scene_tree;
    {
        std::vector<uint32_t> handles(desc.n_objects);
        std::iota(handles.begin(), handles.end(), 0u);
        Builder b(ob, scene_tree);
        scene_tree.root = b.build(handles, 0, handles.size());
        if (b.nan_seen) { err = "NaN object centroid: partial_cmp().unwrap() panics (bvh.rs:99,109,119)"; return RTR_ERR_NAN_CENTROID; }
    }

    // Leaf slots: every object owns a block of leaf records, the blocks laid out in the reference scene tree's
    // depth-first order, so a leaf's index is its reference rank whichever topology is walked.
    std::vector<uint32_t> obj_leaf_base(desc.n_objects, 0u);
    std::vector<uint32_t> vol_leaf_base(desc.n_volumes, 0u);
    {
        uint32_t next = 0;
        std::vector<int32_t> todo{scene_tree.root};
        while (!todo.empty()) { // explicit depth-first walk, left before right
            int32_t ni = todo.back();
            todo.pop_back();
            const TreeNode &n = scene_tree.nodes[ni];
            if (n.kind == T_LEAF) {
                obj_leaf_base[n.handle] = next;
                const rtr_object &o = desc.objects[n.handle];
                next += o.kind == RTR_OBJ_MESH ? desc.meshes[o.index].n_tris : 1u;
            } else if (n.kind == T_SPLIT) {
                todo.push_back(n.right);
                todo.push_back(n.left);
            }
        }
        // the triangles of meshes that bound a Volume: behind the scene tree's leaves (their ranks order ties inside the
        // boundary mesh only)
        for (uint32_t vi = 0; vi < desc.n_volumes; vi++)
            if (volume_is_object[vi] && desc.volumes[vi].boundary_kind == RTR_OBJ_MESH && desc.volumes[vi].boundary_mesh < desc.n_meshes) {
                vol_leaf_base[vi] = next;
                next += desc.meshes[desc.volumes[vi].boundary_mesh].n_tris;
            }
        out.leaf_hot.assign(next, DevLeafHot{});
        out.leaf_cold.assign(next, DevLeafCold{});
    }
    // Optional SAH topology for the scene-level tree (BuildOptions::sah_scene_tree): only when no Volume needs the
    // reference's visit order and every object box has a positive extent on every axis (a zero-thickness union box
    // never passes AABB::hit -- a quirk of the reference that regrouping objects would change visibly).
    Tree scene_fast;
    bool use_scene_fast = false;
    if (opts.sah_scene_tree && desc.n_objects > 2 && desc.n_volumes == 0) {
        bool solid = true;
        for (const Box &b : ob)
            for (int k = 0; k < 3; k++)
                if (!(b.mx[k] > b.mn[k])) solid = false;
        if (solid) {
            std::vector<uint32_t> handles(desc.n_objects);
            std::iota(handles.begin(), handles.end(), 0u);
            SahBuilder sb(ob, scene_fast, tree_height(scene_tree, scene_tree.root));
            scene_fast.root = sb.build(handles, 0, handles.size(), 0);
            use_scene_fast = true;
        }
    }
    const Tree &scene_walk = use_scene_fast ? scene_fast : scene_tree;

    // ---- flatten: preorder nodes, leaves ranked in the reference's depth-first order ----
    std::vector<uint32_t> obj_ref(desc.n_objects, REF_NONE);
    std::vector<Box> obj_box(desc.n_objects);
    std::vector<uint32_t> obj_height(desc.n_objects, 0);
    Flattener fl{desc, meshes, scene_walk, out, obj_ref, obj_box, obj_height, obj_leaf_base, vol_leaf_base, nested, inner_density, true, {}};
    fl.leaf_pairs = opts.leaf_pairs;
    Flattener::Emitted root = fl.emit_scene_node(scene_walk.root);
    out.root_ref = root.ref;
    for (int k = 0; k < 3; k++) { out.root_box[k] = root.box.mn[k]; out.root_box[3 + k] = root.box.mx[k]; }
    out.tree_height = root.height;
    out.stack_depth = root.pushes;
    if (out.leaf_hot.size() > REF_INDEX || out.nodes.size() > REF_INDEX) { err = "scene too large"; return RTR_ERR_INVALID_ARGUMENT; }

    // ---- the linear world's views (scene.rs:141-157): objects in Scene.objects order ----
    out.n_objects = desc.n_objects; out.n_spheres = desc.n_spheres; out.n_meshes = desc.n_meshes;
    out.objects.resize(desc.n_objects);
    for (uint32_t i = 0; i < desc.n_objects; i++) {
        DevObject &d = out.objects[i];
        std::memset(&d, 0, sizeof d);
        d.ref = obj_ref[i];
        for (int k = 0; k < 3; k++) { d.box[k] = obj_box[i].mn[k]; d.box[3 + k] = obj_box[i].mx[k]; }
        out.mesh_height = std::max(out.mesh_height, obj_height[i]);
        if (i < fl.obj_pushes.size()) out.mesh_stack_depth = std::max(out.mesh_stack_depth, fl.obj_pushes[i]);
        if (desc.objects[i].kind == RTR_OBJ_MESH) out.n_triangles += desc.meshes[desc.objects[i].index].n_tris;
    }
    if (out.spheres_only) {
        out.lin_spheres.resize(desc.n_objects);
        for (uint32_t i = 0; i < desc.n_objects; i++) {
            const rtr_sphere &s = desc.spheres[desc.objects[i].index];
            DevLinSphere &d = out.lin_spheres[i];
            std::memset(&d, 0, sizeof d);
            d.center[0] = s.center[0]; d.center[1] = s.center[1]; d.center[2] = s.center[2];
            d.radius = s.radius;
            d.radius2 = s.radius * s.radius;
            d.material = s.material;
        }
    }
    return RTR_OK;
}

// Camera::ray_origin, lib/core/camera.rs:37-56 (cgmath operator order; tan on the host, as the reference)
void camera_ray_origin(const rtr_camera &cam, float origin[3], float horizontal[3], float vertical[3], float llc[3]) {
    const float pi = 3.14159265358979323846f;
    float theta = cam.vertical_fov * pi / 180.0f; // degrees_to_radians, math.rs:31-33
    float h = std::tan(theta / 2.0f);
    float viewport_height = 2.0f * h;
    float viewport_width = cam.aspect_ratio * viewport_height;
    float w[3], u[3], v[3], t[3];
    sub3(cam.lookfrom, cam.lookat, t);
    float inv = 1.0f / std::sqrt((t[0] * t[0] + t[1] * t[1]) + t[2] * t[2]);
    for (int k = 0; k < 3; k++) w[k] = t[k] * inv;
    cross3(cam.vertical, w, t);
    inv = 1.0f / std::sqrt((t[0] * t[0] + t[1] * t[1]) + t[2] * t[2]);
    for (int k = 0; k < 3; k++) u[k] = t[k] * inv;
    cross3(w, u, v);
    for (int k = 0; k < 3; k++) {
        origin[k] = cam.lookfrom[k];
        horizontal[k] = viewport_width * u[k];
        vertical[k] = viewport_height * v[k];
        llc[k] = ((cam.lookfrom[k] - horizontal[k] / 2.0f) + vertical[k] / 2.0f) - w[k];
    }
}

} // namespace rtr
