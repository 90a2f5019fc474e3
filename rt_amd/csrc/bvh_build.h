// bvh_build.h — host BVH builder for the flattened scene (replaces bvh::Bvh::build_par, which the
// reference calls at src/hittable.rs:34).  Binned SAH, BVH2 with child boxes in the parent,
// homogeneous leaves of <= MAX_LEAF primitives, child-reference depth <= SAH_MAX_DEPTH.  Threaded (the top of the
// tree by all threads together, the subtrees below side by side; node-format conversions in pieces); the tree and
// the order of the primitives in its leaves do not depend on the number of threads.
#pragma once
#include <cstdint>
#include <functional>
#include <string>
#include <vector>

#include "rt_layout.h"

namespace rt {

struct PrimBox {
    double lo[3], hi[3];
};

struct BvhResult {
    std::vector<Node64> nodes;          // nodes[0] is the root when n_prims > 0
    std::vector<uint32_t> tri_order;    // leaf order -> index into the input triangle array
    std::vector<uint32_t> sph_order;    // leaf order -> index into the input sphere array
    uint32_t depth = 0;                 // deepest child reference
    uint32_t max_leaf = 0;
};

// fn(k) once for every k in [0, n_pieces), on the build threads (RT_HIP_BUILD_THREADS, default min(cores, 16)) and the
// caller; returns when all calls have returned.  The builder and the flattening loops of rt_scene_create share it.
void for_each_piece(size_t n_pieces, const std::function<void(size_t)>& fn);

// boxes: first n_spheres entries are spheres, the rest triangles (the caller's primitive order).
void build_bvh(const std::vector<PrimBox>& boxes, size_t n_spheres, BvhResult& out);

// Two-level build for scenes made of many separate objects: group_of[p] (one entry per primitive; 0 = loose, g > 0 =
// triangle p belongs to group g of 1 .. n_groups, each of at least 8 triangles) gives every group a tree of its own under a
// top tree over the loose primitives and the groups' boxes.  The result has the layout of build_bvh's.
void build_bvh_grouped(const std::vector<PrimBox>& boxes, size_t n_spheres, const std::vector<uint32_t>& group_of, uint32_t n_groups,
                       BvhResult& out);

// The same result from the device (bvh_gpu.hip: Morton order, Karras radix tree, bottom-up fit; leaf collapse and
// depth-first renumbering on the host).  Returns 0 = ok; > 0 = not applicable (tiny scene, tree deeper than
// BVH_MAX_DEPTH): use build_bvh; < 0 = error, message in err.  Needs a current HIP device.
// phase_s (optional): seconds spent in {upload + kernels + download, host finish}
int build_bvh_device(const std::vector<PrimBox>& boxes, size_t n_spheres, BvhResult& out, std::string& err, double* phase_s = nullptr);

// f32 copy of the nodes with boxes rounded outward.
void narrow_nodes(const std::vector<Node64>& in, std::vector<Node32>& out);

// 16-bit global-grid copy of the nodes.  Returns false (out untouched) when the grid is too coarse
// for the scene: mean surface-area inflation of the leaf boxes above `max_inflation`.
bool quantize_nodes(const std::vector<Node32>& in, std::vector<Node16>& out, QuantGrid& grid, double max_inflation);

// BVH4 over the same leaves: the BVH2 nodes that become BVH4 nodes are chosen for the least summed node area
// (dynamic programme, Wide::make_plan).  Boxes are quantised outward on `grid` (the one quantize_nodes chose).  stack_need =
// entries a near-first traversal that pushes every other hit child can hold at once, worst case.
// Returns false when a box cannot be represented on the grid.
bool widen_nodes(const std::vector<Node64>& in, const QuantGrid& grid, std::vector<Node4>& out, uint32_t& stack_need);
// the same with f32 boxes rounded outward (no grid needed; the caller has checked that coordinates fit f32)
bool widen_nodes_f32(const std::vector<Node64>& in, std::vector<Node4f>& out, uint32_t& stack_need);

// BVH8 over the same leaves (rt_layout.h: Node8): every node absorbs the grandchildren with the largest boxes until it
// has eight children; slots by octant order; boxes on the node's own 8-bit grid, rounded outward and checked exactly.
// The result is the layout of the walk's record pool: record 0 is the root node, a node's internal children and then
// its leaf primitives sit in one block, blocks in depth-first order.
//   node_rec / nodes : nodes[i] belongs at record node_rec[i]
//   tri_rec[k] / sph_rec[k] : record of the k-th triangle / sphere in the BVH2's leaf order (a leaf's primitives stay in a row)
//   n_records : records of the pool; depth : levels of internal nodes (the walk's stack need)
// Returns false when a box cannot be represented (coordinates beyond the exponent range) or the tree is deeper than
// BVH8_MAX_DEPTH: the caller then walks another format.
struct Wide8 {
    std::vector<Node8> nodes;
    std::vector<uint32_t> node_rec;
    std::vector<uint32_t> tri_rec, sph_rec;
    size_t n_records = 0;
    uint32_t depth = 0;
};
bool widen_nodes8(const std::vector<Node64>& in, size_t n_triangles, size_t n_spheres, Wide8& out);

}  // namespace rt
