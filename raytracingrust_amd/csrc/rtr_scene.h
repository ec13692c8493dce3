// rtr_scene.h -- host side of rtr_upload_scene: validate the POD scene, build the reference's two-level
// median-split tree (lib/core/bvh.rs:59-144 over Scene, scene.rs:117-139, and over each Mesh,
// mesh.rs:200-232) and flatten it into the arrays of rtr_device.h.  Pure C++ (no HIP): also used by
// host-only contexts and the CPU test suite.
#pragma once
#include <stdint.h>

#include <string>
#include <vector>

#include "../../include/rtr_hip.h"
#include "rtr_device.h"

namespace rtr {

// A mesh subtree the device builds (BuildOptions::lbvh_mesh_trees; csrc/rtr_lbvh.hip): the host reserves the mesh's
// n_tris - 1 node records and hands over the per-triangle boxes; rtr_upload_scene runs the build and adds the measured
// tree height to the stack bounds.
struct LbvhJob {
    uint32_t node_base, n_tris, leaf_base;
    uint64_t box_offset; // first triangle's box in FlatScene::lbvh_boxes (6 floats per triangle)
    float bounds_min[3], bounds_max[3]; // union of the triangle boxes: the subtree's root box
};

struct FlatScene {
    std::vector<LbvhJob> lbvh_jobs;
    std::vector<float> lbvh_boxes;
    std::vector<DevNode> nodes;
    std::vector<DevLeafHot> leaf_hot;
    std::vector<DevLeafCold> leaf_cold;
    std::vector<DevLinSphere> lin_spheres;
    std::vector<DevObject> objects;
    std::vector<DevMaterial> materials;
    std::vector<float> sky; // float4 texels
    uint32_t sky_w = 0, sky_h = 0;
    uint32_t bg_kind = 0;
    float bg_a[3] = {0, 0, 0}, bg_b[3] = {0, 0, 0};
    uint32_t root_ref = REF_NONE;
    float root_box[6] = {0, 0, 0, 0, 0, 0};
    uint32_t split_nodes = 0;   // Split nodes of the flattened trees, collapsed leaf pairs included
    uint32_t tree_height = 0;   // Split nodes on the longest root-to-leaf path of the scene tree
    uint32_t stack_depth = 0;   // node records on the longest path of the flattened tree: the most the walk can push
                                // (<= tree_height: a collapsed leaf pair has no record)
    uint32_t mesh_stack_depth = 0; // same, over the mesh subtrees alone
    uint32_t mesh_height = 0;   // same, over the mesh subtrees alone (linear world with meshes)
    bool spheres_only = true;
    bool has_volumes = false;   // a Volume leaf exists: traversal must keep the reference's visit set and order
    bool has_mesh_volumes = false; // ... and one of them is bounded by a mesh (nested mesh walk inside Volume::hit)
    bool has_nested_volumes = false; // ... and one of them is bounded by another Volume (one level: the kernels with the nested mesh walk evaluate it)
    uint32_t n_objects = 0, n_spheres = 0, n_meshes = 0, n_triangles = 0;
};

struct BuildOptions {
    // false (default): every mesh subtree is the reference's median-split tree (Mesh::build_bvh, mesh.rs:119-121) --
    // the parity configuration.  true: mesh subtrees are rebuilt with a binned surface-area heuristic over the same
    // triangle boxes: same closest hits (ties still broken by reference rank), about half the traversal work on
    // house.obj; frames can differ from the reference tree's in the ulp-level box-boundary cases of DESIGN.md section 3.
    bool sah_mesh_trees = false;
    // the same for the scene-level tree over the objects (used only for scenes without Volume objects whose object
    // boxes all have positive extent; otherwise the reference tree is kept)
    bool sah_scene_tree = false;
    // Mesh subtrees are built ON THE DEVICE as linear BVHs (Morton sort + Karras radix tree, csrc/rtr_lbvh.hip) over the
    // same triangle boxes; the host skips its O(n log^2 n) reference build for them.  Same closest hits; exact-t ties
    // resolve by triangle handle (the reference's depth-first rank is not computed); needs a device-bound context.
    bool lbvh_mesh_trees = false;
    // A Split whose children are two leaves of one type with consecutive ranks is stored in its parent as one
    // REF_PAIR ref (no node record): the same box test, the same two leaf tests in the same order, one
    // traversal step fewer.  Decisions are unchanged, so this stays on in the parity configuration.
    bool leaf_pairs = true;
};

// Returns RTR_OK or a negative rtr_status; on failure `err` holds the message.
int build_flat_scene(const rtr_scene_desc &desc, FlatScene &out, std::string &err, const BuildOptions &opts = BuildOptions());

// RayOrigin of Camera::ray_origin (lib/core/camera.rs:37-56): origin, horizontal, vertical, lower_left_corner
void camera_ray_origin(const rtr_camera &cam, float origin[3], float horizontal[3], float vertical[3], float llc[3]);

} // namespace rtr
