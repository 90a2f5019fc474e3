// rt_layout.h — HBM data layout of a flattened scene and of the wavefront path state.
// Shared by the host flattening code, the BVH builder and the kernels (all part of librt_hip.so).
#pragma once
#include <stddef.h>
#include <stdint.h>

#include "../../include/rt_hip.h"

namespace rt {

// ---- BVH child references ------------------------------------------------------------------
// internal node : index into nodes[]                         (bit 31 clear)
// leaf          : bit31 | sphere<<30 | (count-1)<<28 | first  (first indexes tris[] or sphs[], leaf order)
// empty         : 0xFFFFFFFF
constexpr uint32_t REF_EMPTY = 0xFFFFFFFFu;
constexpr uint32_t REF_LEAF = 0x80000000u;
constexpr uint32_t REF_SPHERE = 0x40000000u;
constexpr uint32_t REF_FIRST_MASK = 0x0FFFFFFFu;
constexpr int MAX_LEAF = 4;
// Traversal keeps one pending entry per ancestor, so the per-lane LDS stack needs `depth` entries;
// the kernels provide stacks for child references down to depth BVH_MAX_DEPTH.  The host SAH builder keeps its
// trees within SAH_MAX_DEPTH (median-split fallback); Morton-order trees from the device builder may be deeper.
constexpr int BVH_MAX_DEPTH = 64;
constexpr int SAH_MAX_DEPTH = 32;

// BVH2 node with both child boxes in the parent, f64 bounds: one 128-byte line per node.
struct alignas(128) Node64 {
    double lo_l[3], hi_l[3];
    double lo_r[3], hi_r[3];
    uint32_t left, right;
    uint32_t _pad[6];
};
static_assert(sizeof(Node64) == 128, "Node64 layout");

// BVH2 node with conservative f32 child boxes (rounded outward from f64): 64 bytes, two per line.
struct alignas(64) Node32 {
    float lo_l[3], hi_l[3];
    float lo_r[3], hi_r[3];
    uint32_t left, right;
    uint32_t _pad[2];
};
static_assert(sizeof(Node32) == 64, "Node32 layout");

// BVH2 node with child boxes quantised (outward) to a global 16-bit grid over the scene bounds:
// 32 bytes, four per line.  Decoded as fmaf(q, cell, origin) in f32; the builder verifies that
// every decoded bound encloses the f32 bound it replaces.  Used for incoherent rays, whose
// per-lane node fetches are bound by the vector-memory address unit (bytes per lane per step).
struct alignas(32) Node16 {
    uint16_t lo_l[3], hi_l[3];
    uint16_t lo_r[3], hi_r[3];
    uint32_t left, right;
};
static_assert(sizeof(Node16) == 32, "Node16 layout");
struct QuantGrid {
    float origin[3];
    float cell[3];
};

// BVH4 node on the same 16-bit grid: the boxes and references of up to four children in half a
// 128-byte line.  Built by collapsing the BVH2 (bvh_build.cpp: widen_nodes).  Incoherent rays pay per
// node FETCH (one L2 line request per lane, tools/micro/gather_bench.hip) and per dependent step, not
// per byte, so halving the visits per ray is worth more than the bytes.  Unused slots come last and
// hold REF_EMPTY.
struct alignas(64) Node4 {
    uint16_t box[4][6];      // child k: lo x, y, z, hi x, y, z
    uint32_t child[4];
};
static_assert(sizeof(Node4) == 64, "Node4 layout");
constexpr int BVH4_MAX_STACK = 64;   // widest per-lane stack a kernel instance provides

// The same BVH4 with f32 boxes (rounded outward) for scenes whose 16-bit grid would be too coarse (a huge
// ground plane next to small spheres): one 128-byte line per node.
struct alignas(128) Node4f {
    float box[4][6];         // child k: lo x, y, z, hi x, y, z
    uint32_t child[4];
    uint32_t _pad[4];
};
static_assert(sizeof(Node4f) == 128, "Node4f layout");

// Primitive records: 80 bytes each, triangles and spheres alike, so that one array — and, for the BVH8 walk, one pool
// of 80-byte records holding nodes AND primitives — is addressed with one stride.  `kind` (the last word of either
// record) says which it is: the BVH8 walk reads it from the record it fetched, the other walks know it from the
// leaf reference.
enum { PRIM_KIND_TRIANGLE = 0, PRIM_KIND_SPHERE = 1 };
// Triangle geometry in leaf order: a, e1 = b - a, e2 = c - a (src/hittable.rs:412-413), f64.
struct alignas(16) TriGeom {
    double a[3], e1[3], e2[3];
    uint32_t prim;      // index in the caller's [spheres..., triangles...] order
    uint32_t kind;      // PRIM_KIND_TRIANGLE
};
static_assert(sizeof(TriGeom) == 80, "TriGeom layout");

// Sphere geometry in leaf order; rot2 = third row of the texture rotation (NaN-uv rejection test).
struct alignas(16) SphGeom {
    double c[3], r;
    double rot2[3];
    uint32_t prim;
    uint32_t _pad[4];
    uint32_t kind;      // PRIM_KIND_SPHERE (same offset as TriGeom::kind)
};
static_assert(sizeof(SphGeom) == 80 && offsetof(SphGeom, kind) == offsetof(TriGeom, kind), "SphGeom layout");
constexpr uint32_t PRIM_RECORD_BYTES = 80;

// BVH8 node for incoherent rays (scenes of any size; the format of choice beyond the caches, where a walk is bound by
// dependent memory round trips): one 80-byte record of the pool [nodes and primitives interleaved], after Ylitie,
// Karras, Laine, "Efficient Incoherent Ray Traversal on GPUs Through Compressed Wide BVHs" (HPG 2017) — re-designed for
// this walk (one fetch per step, f64 primitive tests, leaves of up to four primitives):
//   - the boxes of up to eight children as 8-bit planes on a grid LOCAL to the node: plane = origin[k] + q * 2^(ex[k]-127),
//     origin = the node box's lower corner rounded down to f32, the exponent the smallest that reaches the upper corner
//     with q <= 255; the builder checks in 64-bit-mantissa arithmetic that these planes enclose every child box.
//   - ONE base: the node's internal children are the records base, base+1, ... (in slot order), its leaf children's
//     primitives the records right behind them (slot order, a leaf's primitives in a row).  inib has nibble s = 0xF when
//     slot s is an internal child; pmask has bit 4s+j set when slot s is a leaf with more than j primitives.
//   - slots are assigned so that visiting them in ascending order of (slot XOR octant of the ray direction) is roughly
//     front to back: no sort in the kernel.
struct alignas(16) Node8 {
    float origin[3];
    uint8_t ex[3];           // biased exponents of the cell sizes (uint_as_float(ex << 23) = 2^(ex-127))
    uint8_t _pad0;
    uint32_t base;           // record index of the first internal child
    uint32_t inib;           // internal-child slots, one nibble each
    uint32_t pmask;          // primitives of the leaf slots, one nibble each
    uint32_t _pad1;
    uint8_t qlo[3][8];       // [axis][slot]
    uint8_t qhi[3][8];
};
static_assert(sizeof(Node8) == 80, "Node8 layout");
constexpr int BVH8_MAX_DEPTH = 24;    // levels of internal nodes a BVH8 walk kernel provides stack for (two words each)

// Shading records, indexed by primitive id (original order).
struct alignas(16) TriShade {
    double n[3];
    double uv_left, uv_width, uv_bot, uv_height;   // bbox of the three vertex uvs (src/hittable.rs:469-476)
    uint32_t material;
    uint32_t _pad;
};
static_assert(sizeof(TriShade) == 64, "TriShade layout");

struct alignas(16) SphShade {
    double c[3], r;
    double rot[9];
    uint32_t material;
    uint32_t _pad;
};
static_assert(sizeof(SphShade) == 112, "SphShade layout");

// Material with the root texture node folded in: a solid-colour material (the common case) is
// shaded from this one 64-byte record without touching the texture table.
struct alignas(16) ShadeMat {
    uint32_t kind, has_fuzz;
    uint32_t texture, tex_kind;     // root texture node and its kind
    double fuzz, ior;
    double rgb[3];                  // root colour when tex_kind == RT_TEX_SOLID
    uint32_t _pad[2];
};
static_assert(sizeof(ShadeMat) == 64, "ShadeMat layout");

struct DScene {
    const Node64* nodes64;
    const Node32* nodes32;
    const Node16* nodes16;       // optional (null when quantisation would inflate the leaves too much)
    const Node4* nodes4;         // optional: BVH4 over the same leaves, same grid (per-lane traversal prefers it)
    const Node4f* nodes4f;       // optional: BVH4 with f32 boxes, when there is no grid
    const char* pool8;           // optional: the BVH8 walk's pool of 80-byte records (Node8, TriGeom, SphGeom); record 0 = root.
                                 // tris and sphs then both point at it and the BVH2 arrays' leaf references count its records.
    uint32_t depth8;             // levels of internal BVH8 nodes below the root = (base, pending) pairs its walk can hold at once
    const TriGeom* tris;
    const SphGeom* sphs;
    const TriShade* tri_shade;   // [n_triangles], index = prim - n_spheres
    const SphShade* sph_shade;   // [n_spheres]
    const ShadeMat* mats;
    const rt_texture* texs;
    const uint8_t* texels;
    uint32_t n_spheres, n_prims;
    uint32_t root;               // REF_EMPTY for an empty scene, else 0
    uint32_t bvh_depth;          // deepest child reference = traversal stack entries needed
    QuantGrid grid;              // of nodes16 and nodes4
    uint32_t stack4;             // traversal stack entries nodes4 / nodes4f need in the worst case
    uint32_t all_lanes_load;     // BVH4 walks: lanes without work fetch node 0 instead of sitting out the loads (see walk_body)
    // BVH4 walks address ONE pool [nodes4 or nodes4f | tris | sphs]: the low 28 bits of a child reference inside nodes4 /
    // nodes4f are the record's offset in the pool in 16-byte units (node, triangle or sphere alike), so a step's fetch
    // address is base + (ref & mask) * 16 whatever the lane is at.  nodes4 / nodes4f, tris and sphs point into it; the
    // references of the BVH2 node arrays keep counting records.
    const char* pool4;
    double grid_inv_cell[3];     // 1 / grid.cell (f64): a ray is moved into cell units with a multiplication
    rt_sky sky;
};

// ---- wavefront path state (SoA, one slot per in-flight path) ---------------------------------
struct RayQueue {
    double* o[3];
    double* d[3];
    double* thr[3];     // path throughput
    uint32_t* pid;      // path id inside the pass: local_sample * n_pixels + pixel
};
// A ray queue is stored as NSHARD independently appended segments so that compaction spreads its
// (returning) atomics over NSHARD counters: one counter saturates at ~88 appends/us on MI355X, which
// at one append per wave cost more than the shading itself.  Producer wave w appends to shard
// w % NSHARD; consumer wave b reads shard b % NSHARD at offset 64*(b / NSHARD), which interleaves
// the shards back into (roughly) the producer's order — camera tiles stay neighbours in flight, so
// the next bounce still sweeps the scene once instead of NSHARD times.
//
// The geometry is FIXED per pass: shard k owns the physical slots [k * shard_cap, (k + 1) * shard_cap), with
// shard_cap = 64 * ceil(waves of the pass / NSHARD) — the survivors of the input waves w = k (mod NSHARD) always fit.
// The only thing that changes from bounce to bounce are the NSHARD entry counts, and they stay in DEVICE memory
// (written by the producer's atomics, read by every consumer wave): the host never sees them, so it can queue the
// kernels of a whole pass without waiting for any of them.
constexpr int NSHARD = 64;
constexpr uint32_t COUNT_STRIDE = 32;            // u32 between two shard counters (128 B apart)
constexpr uint32_t QUEUE_NWAVES_WORD = 1;        // count[1] (a gap word): the queue's logical waves, written by k_walk for k_shade
struct QueueGeom {
    const uint32_t* count;      // device: count[COUNT_STRIDE * k] = live entries of shard k; nullptr = dense queue (below)
    uint32_t shard_cap;         // physical slots per shard (a multiple of 64)
    uint32_t n_direct;          // count == nullptr: the queue holds entries 0 .. n_direct-1, entry p in logical wave p / 64
};
// slots per shard for a pass of n paths (host and device)
static inline uint32_t shard_cap_for(uint64_t n) {
    const uint64_t waves = (n + 63u) / 64u;
    return (uint32_t)(64u * ((waves + NSHARD - 1) / NSHARD));
}

struct HitBuf {
    double* t;
    double* bu;         // triangle barycentrics u, v of the accepted hit
    double* bv;
    uint32_t* prim;     // RT_MISS = escaped to the sky
};

}  // namespace rt
