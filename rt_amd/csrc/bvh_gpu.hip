// bvh_gpu.hip — BVH construction on the device: replaces the host build that stands
// where the reference calls bvh::Bvh::build_par (src/hittable.rs:34) when RT_HIP_BUILDER=lbvh.
//
//   1. k_morton        63-bit Morton code of every primitive's box centre (21 bits per axis, f64 -> grid)
//   2. rocprim         radix sort of (code, primitive) pairs
//   3. k_radix_tree    Karras' binary radix tree over the sorted codes (ties broken by position): one thread per
//                      internal node finds its range and split; n - 1 internal nodes, node 0 is the root
//   4. k_fit           bottom-up: one thread per leaf climbs; the second thread to reach a node (device-scope
//                      counter) writes both child boxes into it (Node64: children's boxes in the parent)
//   host finish        subtrees of <= RT_LEAF_MAX primitives of one kind become leaves, nodes are renumbered in
//                      depth-first order (the layout bvh_build.cpp produces), depth and leaf sizes are recorded
//
// A Morton-order tree is built in milliseconds but its boxes overlap more than those of the binned-SAH build
// (more node visits per ray), so it is opt-in: time-to-first-pixel on very large scenes.  Any conservative BVH
// gives the same closest hits; the parity tests run on both builders.
#include <hip/hip_runtime.h>

#include <chrono>
#include <cmath>
#include <cstring>
#include <string.h>

#include <rocprim/rocprim.hpp>

#include "bvh_build.h"

#ifndef RT_LEAF_MAX
#define RT_LEAF_MAX MAX_LEAF
#endif

namespace rt {
namespace {

#define GCHK(x)                                                        \
    do {                                                               \
        hipError_t e_ = (x);                                           \
        if (e_ != hipSuccess) { err = hipGetErrorString(e_); goto fail; } \
    } while (0)

constexpr uint32_t LEAF_BIT = 0x80000000u;
constexpr int GB = 256;

__device__ __forceinline__ uint64_t spread21(uint32_t v) {
    uint64_t x = v & 0x1FFFFFu;
    x = (x | x << 32) & 0x1F00000000FFFFull;
    x = (x | x << 16) & 0x1F0000FF0000FFull;
    x = (x | x << 8) & 0x100F00F00F00F00Full;
    x = (x | x << 4) & 0x10C30C30C30C30C3ull;
    x = (x | x << 2) & 0x1249249249249249ull;
    return x;
}

struct Grid3 { double lo[3], scale[3]; };

__global__ __launch_bounds__(GB) void k_morton(const PrimBox* boxes, uint32_t n, Grid3 g, uint64_t* keys, uint32_t* vals) {
    const uint32_t i = blockIdx.x * GB + threadIdx.x;
    if (i >= n) return;
    uint64_t code = 0;
    for (int k = 0; k < 3; ++k) {
        const double c = 0.5 * (boxes[i].lo[k] + boxes[i].hi[k]);
        double q = (c - g.lo[k]) * g.scale[k];
        q = fmin(fmax(q, 0.0), 2097151.0);
        code |= spread21((uint32_t)q) << k;
    }
    keys[i] = code;
    vals[i] = i;
}

// length of the common prefix of the keys at sorted positions i and j; equal keys fall back to the positions
__device__ __forceinline__ int delta(const uint64_t* keys, int n, int i, int j) {
    if (j < 0 || j >= n) return -1;
    const uint64_t a = keys[i], b = keys[j];
    if (a != b) return __clzll((long long)(a ^ b));
    return 64 + __clz(i ^ j);
}

// Karras 2012, one thread per internal node.  child: index of an internal node, or LEAF_BIT | sorted position.
__global__ __launch_bounds__(GB) void k_radix_tree(const uint64_t* keys, int n, uint32_t* left, uint32_t* right, uint32_t* first,
                                                   uint32_t* parent_internal, uint32_t* parent_leaf) {
    const int i = blockIdx.x * GB + threadIdx.x;
    if (i >= n - 1) return;
    const int d = delta(keys, n, i, i + 1) - delta(keys, n, i, i - 1) >= 0 ? 1 : -1;
    const int dmin = delta(keys, n, i, i - d);
    int lmax = 2;
    while (delta(keys, n, i, i + lmax * d) > dmin) lmax <<= 1;
    int l = 0;
    for (int t = lmax >> 1; t >= 1; t >>= 1)
        if (delta(keys, n, i, i + (l + t) * d) > dmin) l += t;
    const int j = i + l * d;
    const int dnode = delta(keys, n, i, j);
    int s = 0;
    for (int t = (l + 1) >> 1;; t = (t + 1) >> 1) {
        if (delta(keys, n, i, i + (s + t) * d) > dnode) s += t;
        if (t <= 1) break;
    }
    const int gamma = i + s * d + (d < 0 ? -1 : 0);
    const int lo = i < j ? i : j, hi = i < j ? j : i;
    const uint32_t L = lo == gamma ? (LEAF_BIT | (uint32_t)gamma) : (uint32_t)gamma;
    const uint32_t R = hi == gamma + 1 ? (LEAF_BIT | (uint32_t)(gamma + 1)) : (uint32_t)(gamma + 1);
    left[i] = L;
    right[i] = R;
    first[i] = (uint32_t)lo;
    if (L & LEAF_BIT) parent_leaf[gamma] = (uint32_t)i; else parent_internal[gamma] = (uint32_t)i;
    if (R & LEAF_BIT) parent_leaf[gamma + 1] = (uint32_t)i; else parent_internal[gamma + 1] = (uint32_t)i;
}

struct SubInfo { double lo[3], hi[3]; uint32_t size, kind; };   // kind: bit 0 = holds triangles, bit 1 = holds spheres

__device__ __forceinline__ double aload(const double* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ uint32_t aload(const uint32_t* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

__device__ __forceinline__ void child_info(uint32_t ref, const PrimBox* boxes, const uint32_t* vals, uint32_t n_spheres,
                                           const SubInfo* sub, SubInfo& o) {
    if (ref & LEAF_BIT) {
        const uint32_t p = vals[ref & ~LEAF_BIT];
        for (int k = 0; k < 3; ++k) { o.lo[k] = boxes[p].lo[k]; o.hi[k] = boxes[p].hi[k]; }
        o.size = 1;
        o.kind = p < n_spheres ? 2u : 1u;
    } else {   // written by another thread, possibly on another XCD: device-scope loads after the acquire fence
        const SubInfo* s = &sub[ref];
        for (int k = 0; k < 3; ++k) { o.lo[k] = aload(&s->lo[k]); o.hi[k] = aload(&s->hi[k]); }
        o.size = aload(&s->size);
        o.kind = aload(&s->kind);
    }
}

// One thread per leaf climbs towards the root; at every node the first arrival stops, the second (which then knows
// both subtrees are finished) fills the node.  No thread waits for another: the kernel cannot hang.
__global__ __launch_bounds__(GB) void k_fit(const PrimBox* boxes, const uint32_t* vals, uint32_t n, uint32_t n_spheres,
                                            const uint32_t* left, const uint32_t* right, const uint32_t* parent_internal,
                                            const uint32_t* parent_leaf, uint32_t* arrivals, SubInfo* sub, Node64* nodes) {
    const uint32_t i = blockIdx.x * GB + threadIdx.x;
    if (i >= n) return;
    uint32_t cur = parent_leaf[i];
    for (;;) {
        __threadfence();                                   // release what this thread wrote below
        if (atomicAdd(&arrivals[cur], 1u) == 0u) return;
        __threadfence();                                   // acquire the sibling subtree
        const uint32_t L = left[cur], R = right[cur];
        SubInfo a, b;
        child_info(L, boxes, vals, n_spheres, sub, a);
        child_info(R, boxes, vals, n_spheres, sub, b);
        Node64& nd = nodes[cur];
        SubInfo me;
        for (int k = 0; k < 3; ++k) {
            nd.lo_l[k] = a.lo[k]; nd.hi_l[k] = a.hi[k];
            nd.lo_r[k] = b.lo[k]; nd.hi_r[k] = b.hi[k];
            me.lo[k] = fmin(a.lo[k], b.lo[k]);
            me.hi[k] = fmax(a.hi[k], b.hi[k]);
        }
        nd.left = L; nd.right = R;
        nd._pad[0] = a.size; nd._pad[1] = b.size; nd._pad[2] = a.kind; nd._pad[3] = b.kind; nd._pad[4] = 0; nd._pad[5] = 0;
        me.size = a.size + b.size;
        me.kind = a.kind | b.kind;
        sub[cur] = me;
        if (cur == 0) return;
        cur = parent_internal[cur];
    }
}

}  // namespace

int build_bvh_device(const std::vector<PrimBox>& boxes, size_t n_spheres, BvhResult& out, std::string& err, double* phase_s) {
    using clk = std::chrono::steady_clock;
    const clk::time_point t_start = clk::now();
    out = BvhResult();
    const size_t n = boxes.size();
    if (n < 64 || n >= (size_t(1) << 28)) return 1;        // tiny scenes: the host builder (single-leaf cases live there)
    Grid3 g;
    for (int k = 0; k < 3; ++k) {
        double lo = INFINITY, hi = -INFINITY;
        for (const PrimBox& b : boxes) {
            const double c = 0.5 * (b.lo[k] + b.hi[k]);
            lo = std::fmin(lo, c); hi = std::fmax(hi, c);
        }
        if (!std::isfinite(lo) || !std::isfinite(hi)) return 1;
        g.lo[k] = lo;
        g.scale[k] = hi > lo ? 2097152.0 / (hi - lo) : 0.0;
    }
    const uint32_t N = (uint32_t)n;
    PrimBox* d_boxes = nullptr;
    uint64_t *d_k0 = nullptr, *d_k1 = nullptr;
    uint32_t *d_v0 = nullptr, *d_v1 = nullptr, *d_left = nullptr, *d_right = nullptr, *d_first = nullptr, *d_pi = nullptr,
             *d_pl = nullptr, *d_arr = nullptr;
    SubInfo* d_sub = nullptr;
    Node64* d_nodes = nullptr;
    void* d_tmp = nullptr;
    size_t tmp_bytes = 0;
    std::vector<uint32_t> vals, first;
    std::vector<Node64> raw;
    int rc = -1;
    {
        GCHK(hipMalloc(&d_boxes, n * sizeof(PrimBox)));
        GCHK(hipMalloc(&d_k0, n * 8)); GCHK(hipMalloc(&d_k1, n * 8));
        GCHK(hipMalloc(&d_v0, n * 4)); GCHK(hipMalloc(&d_v1, n * 4));
        GCHK(hipMalloc(&d_left, n * 4)); GCHK(hipMalloc(&d_right, n * 4)); GCHK(hipMalloc(&d_first, n * 4));
        GCHK(hipMalloc(&d_pi, n * 4)); GCHK(hipMalloc(&d_pl, n * 4)); GCHK(hipMalloc(&d_arr, n * 4));
        GCHK(hipMalloc(&d_sub, n * sizeof(SubInfo)));
        GCHK(hipMalloc(&d_nodes, n * sizeof(Node64)));
        GCHK(hipMemcpy(d_boxes, boxes.data(), n * sizeof(PrimBox), hipMemcpyHostToDevice));
        const unsigned grid = (unsigned)((n + GB - 1) / GB);
        hipLaunchKernelGGL(k_morton, dim3(grid), dim3(GB), 0, 0, d_boxes, N, g, d_k0, d_v0);
        GCHK(rocprim::radix_sort_pairs(nullptr, tmp_bytes, d_k0, d_k1, d_v0, d_v1, n, 0, 63, (hipStream_t)0));
        GCHK(hipMalloc(&d_tmp, tmp_bytes));
        GCHK(rocprim::radix_sort_pairs(d_tmp, tmp_bytes, d_k0, d_k1, d_v0, d_v1, n, 0, 63, (hipStream_t)0));
        GCHK(hipMemsetAsync(d_arr, 0, n * 4, 0));
        GCHK(hipMemsetAsync(d_pi, 0, n * 4, 0));
        hipLaunchKernelGGL(k_radix_tree, dim3(grid), dim3(GB), 0, 0, d_k1, (int)N, d_left, d_right, d_first, d_pi, d_pl);
        hipLaunchKernelGGL(k_fit, dim3(grid), dim3(GB), 0, 0, d_boxes, d_v1, N, (uint32_t)n_spheres, d_left, d_right, d_pi, d_pl, d_arr,
                           d_sub, d_nodes);
        GCHK(hipGetLastError());
        vals.resize(n); first.resize(n); raw.resize(n - 1);
        GCHK(hipMemcpy(vals.data(), d_v1, n * 4, hipMemcpyDeviceToHost));
        GCHK(hipMemcpy(first.data(), d_first, n * 4, hipMemcpyDeviceToHost));
        GCHK(hipMemcpy(raw.data(), d_nodes, (n - 1) * sizeof(Node64), hipMemcpyDeviceToHost));
        rc = 0;
    }
fail:
    (void)hipFree(d_boxes); (void)hipFree(d_k0); (void)hipFree(d_k1); (void)hipFree(d_v0); (void)hipFree(d_v1);
    (void)hipFree(d_left); (void)hipFree(d_right); (void)hipFree(d_first); (void)hipFree(d_pi); (void)hipFree(d_pl);
    (void)hipFree(d_arr); (void)hipFree(d_sub); (void)hipFree(d_nodes); (void)hipFree(d_tmp);
    if (rc) return rc;
    const clk::time_point t_device = clk::now();

    // ---- host finish: leaf order = sorted order per kind; collapse; depth-first renumbering
    std::vector<uint32_t> rank(n + 1);                     // primitives of the same kind before sorted position p
    {
        uint32_t ns = 0, nt = 0;
        out.sph_order.reserve(n_spheres);
        out.tri_order.reserve(n - n_spheres);
        for (size_t p = 0; p < n; ++p) {
            if (vals[p] < n_spheres) { rank[p] = ns++; out.sph_order.push_back(vals[p]); }
            else { rank[p] = nt++; out.tri_order.push_back(vals[p] - (uint32_t)n_spheres); }
        }
    }
    auto child_ref = [&](uint32_t raw_ref, uint32_t size, uint32_t kind, bool& internal) -> uint32_t {
        internal = false;
        const uint32_t p0 = (raw_ref & LEAF_BIT) ? (raw_ref & ~LEAF_BIT) : first[raw_ref];
        if (size <= (uint32_t)RT_LEAF_MAX && (kind == 1u || kind == 2u)) {
            out.max_leaf = std::max(out.max_leaf, size);
            return REF_LEAF | (kind == 2u ? REF_SPHERE : 0u) | ((size - 1u) << 28) | rank[p0];
        }
        if (raw_ref & LEAF_BIT) { err = "internal: LBVH leaf of mixed kind"; return REF_EMPTY; }
        internal = true;
        return raw_ref;
    };
    struct Item { uint32_t old_index, new_index, depth; };
    std::vector<Item> stack;
    out.nodes.reserve(n / 2 + 16);
    out.nodes.emplace_back();
    stack.push_back({0u, 0u, 0u});
    size_t visited = 0;
    while (!stack.empty()) {
        const Item it = stack.back();
        stack.pop_back();
        if (++visited > n) { err = "internal: LBVH is not a tree"; return -1; }
        const Node64 src = raw[it.old_index];
        Node64 nd = src;
        for (uint32_t& p : nd._pad) p = 0;
        bool il = false, ir = false;
        const uint32_t l = child_ref(src.left, src._pad[0], src._pad[2], il);
        const uint32_t r = child_ref(src.right, src._pad[1], src._pad[3], ir);
        if (l == REF_EMPTY || r == REF_EMPTY) return -1;
        if (src._pad[0] + src._pad[1] == 0) { err = "internal: LBVH node was never fitted"; return -1; }
        out.depth = std::max(out.depth, it.depth + 1);
        nd.left = l; nd.right = r;
        // left subtree first in memory (right is pushed first so that left is numbered next)
        uint32_t nl = 0, nr = 0;
        if (il) { nl = (uint32_t)out.nodes.size(); out.nodes.emplace_back(); nd.left = nl; }
        if (ir) { nr = (uint32_t)out.nodes.size(); out.nodes.emplace_back(); nd.right = nr; }
        if (ir) stack.push_back({r, nr, it.depth + 1});
        if (il) stack.push_back({l, nl, it.depth + 1});
        out.nodes[it.new_index] = nd;
    }
    if (phase_s) {
        phase_s[0] = std::chrono::duration<double>(t_device - t_start).count();
        phase_s[1] = std::chrono::duration<double>(clk::now() - t_device).count();
    }
    if (out.depth > (uint32_t)BVH_MAX_DEPTH) return 2;       // deeper than the traversal stacks: the caller falls back
    return 0;
}

}  // namespace rt
