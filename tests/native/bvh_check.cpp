// Host-only check of the BVH builders' CPU code (bvh_build.cpp): builds trees over random boxes — sequentially and
// with worker threads — and validates them: every primitive in exactly one leaf, leaves homogeneous and <= MAX_LEAF,
// every child box encloses its subtree, depth within SAH_MAX_DEPTH, f32 / u16 / BVH4 copies conservative.
// Compiled by tests/test_bvh_host.py with g++ -fsanitize=address,undefined (no GPU, no HIP).
#include <cmath>
#include <cstdio>
#include <cstring>
#include <cstdlib>
#include <random>
#include <vector>

#include "../../rt_amd/csrc/bvh_build.h"

using namespace rt;

struct B3 { double lo[3], hi[3]; };
static int fails = 0;
#define CHECK(c, ...) do { if (!(c)) { if (fails++ < 10) { printf("FAIL %s:%d: ", __FILE__, __LINE__); printf(__VA_ARGS__); printf("\n"); } } } while (0)

static B3 walk(const BvhResult& r, const std::vector<PrimBox>& boxes, size_t n_spheres, uint32_t ref, int depth,
               std::vector<int>& seen, uint32_t& max_depth) {
    B3 b = {{INFINITY, INFINITY, INFINITY}, {-INFINITY, -INFINITY, -INFINITY}};
    max_depth = std::max(max_depth, (uint32_t)depth);
    if (ref & REF_LEAF) {
        const uint32_t first = ref & REF_FIRST_MASK, count = ((ref >> 28) & 3u) + 1u;
        const bool sph = ref & REF_SPHERE;
        CHECK(count <= (uint32_t)MAX_LEAF, "leaf of %u", count);
        for (uint32_t k = 0; k < count; ++k) {
            const size_t order_size = sph ? r.sph_order.size() : r.tri_order.size();
            CHECK(first + k < order_size, "leaf range beyond the order array");
            if (first + k >= order_size) break;
            const uint32_t p = sph ? r.sph_order[first + k] : r.tri_order[first + k] + (uint32_t)n_spheres;
            seen[p]++;
            for (int a = 0; a < 3; ++a) { b.lo[a] = std::fmin(b.lo[a], boxes[p].lo[a]); b.hi[a] = std::fmax(b.hi[a], boxes[p].hi[a]); }
        }
        return b;
    }
    CHECK(ref < r.nodes.size(), "node index %u of %zu", ref, r.nodes.size());
    const Node64& nd = r.nodes[ref];
    B3 l = walk(r, boxes, n_spheres, nd.left, depth + 1, seen, max_depth);
    for (int a = 0; a < 3; ++a) CHECK(nd.lo_l[a] <= l.lo[a] && nd.hi_l[a] >= l.hi[a], "left box of node %u does not enclose", ref);
    b = l;
    if (nd.right != REF_EMPTY) {
        B3 rr = walk(r, boxes, n_spheres, nd.right, depth + 1, seen, max_depth);
        for (int a = 0; a < 3; ++a) {
            CHECK(nd.lo_r[a] <= rr.lo[a] && nd.hi_r[a] >= rr.hi[a], "right box of node %u does not enclose", ref);
            b.lo[a] = std::fmin(b.lo[a], rr.lo[a]); b.hi[a] = std::fmax(b.hi[a], rr.hi[a]);
        }
    }
    return b;
}

// ---- BVH8 (bvh_wide8.cpp): walks the record layout from the root; returns the exact bounds of the subtree.  Checks: a
// node's internal children are the records base, base+1, ... in slot order and its leaves' primitives follow them; the
// planes origin + q * 2^(ex-127) of every child slot enclose that child's primitives (the grid is node-local, so this
// is checked against the true bounds, in 64-bit-mantissa arithmetic); every primitive record is reached exactly once.
struct Pool8View {
    const Wide8& w;
    std::vector<int32_t> node_at;        // record -> node index, -1 otherwise
    std::vector<int32_t> prim_at;        // record -> primitive (caller's order), -1 otherwise
};
static B3 walk8(const Pool8View& v, const std::vector<PrimBox>& boxes, uint32_t rec, uint32_t level, std::vector<int>& seen, uint32_t& max_level,
                size_t& n_nodes) {
    B3 all = {{INFINITY, INFINITY, INFINITY}, {-INFINITY, -INFINITY, -INFINITY}};
    CHECK(rec < v.node_at.size() && v.node_at[rec] >= 0, "record %u is not a node", rec);
    if (rec >= v.node_at.size() || v.node_at[rec] < 0) return all;
    const Node8& nd = v.w.nodes[(size_t)v.node_at[rec]];
    ++n_nodes;
    max_level = std::max(max_level, level);
    uint32_t n_int = 0;
    for (int s = 0; s < 8; ++s) n_int += ((nd.inib >> (4 * s)) & 0xF) ? 1 : 0;
    uint32_t child_rec = nd.base, prim_rec = nd.base + n_int;
    for (int s = 0; s < 8; ++s) {
        const uint32_t in = (nd.inib >> (4 * s)) & 0xF, pm = (nd.pmask >> (4 * s)) & 0xF;
        CHECK(in == 0 || in == 0xF, "inib nibble %x", in);
        CHECK(!(in && pm), "slot %d is both internal and leaf", s);
        CHECK(pm == 0 || pm == 1 || pm == 3 || pm == 7 || pm == 15, "pmask nibble %x", pm);
        if (!in && !pm) continue;
        B3 b = {{INFINITY, INFINITY, INFINITY}, {-INFINITY, -INFINITY, -INFINITY}};
        if (in) {
            b = walk8(v, boxes, child_rec++, level + 1, seen, max_level, n_nodes);
        } else {
            for (uint32_t j = 0; j < 4 && ((pm >> j) & 1); ++j) {
                CHECK(prim_rec < v.prim_at.size() && v.prim_at[prim_rec] >= 0, "record %u is not a primitive", prim_rec);
                if (prim_rec < v.prim_at.size() && v.prim_at[prim_rec] >= 0) {
                    const int p = v.prim_at[prim_rec];
                    seen[p]++;
                    for (int a = 0; a < 3; ++a) { b.lo[a] = std::fmin(b.lo[a], boxes[p].lo[a]); b.hi[a] = std::fmax(b.hi[a], boxes[p].hi[a]); }
                }
                ++prim_rec;
            }
        }
        for (int a = 0; a < 3; ++a) {
            const long double lo = (long double)nd.origin[a] + std::ldexp((long double)nd.qlo[a][s], (int)nd.ex[a] - 127);
            const long double hi = (long double)nd.origin[a] + std::ldexp((long double)nd.qhi[a][s], (int)nd.ex[a] - 127);
            CHECK(lo <= (long double)b.lo[a] && hi >= (long double)b.hi[a], "BVH8 node at record %u slot %d axis %d does not enclose its child", rec, s, a);
            CHECK(nd.ex[a] >= 1 && nd.ex[a] <= 254, "exponent byte %u", nd.ex[a]);
            all.lo[a] = std::fmin(all.lo[a], b.lo[a]); all.hi[a] = std::fmax(all.hi[a], b.hi[a]);
        }
    }
    return all;
}
static void check_wide8(const char* what, const BvhResult& r, const std::vector<PrimBox>& boxes, size_t n_spheres) {
    Wide8 w;
    if (r.nodes.empty()) return;
    const bool ok = widen_nodes8(r.nodes, r.tri_order.size(), r.sph_order.size(), w);
    CHECK(ok, "%s: widen_nodes8 failed", what);
    if (!ok) return;
    CHECK(w.n_records == w.nodes.size() + boxes.size(), "%s: %zu records for %zu nodes + %zu primitives", what, w.n_records, w.nodes.size(), boxes.size());
    Pool8View v{w, std::vector<int32_t>(w.n_records, -1), std::vector<int32_t>(w.n_records, -1)};
    size_t clash = 0;
    for (size_t i = 0; i < w.nodes.size(); ++i) { clash += v.node_at[w.node_rec[i]] >= 0; v.node_at[w.node_rec[i]] = (int32_t)i; }
    for (size_t k = 0; k < w.tri_rec.size(); ++k) { clash += v.node_at[w.tri_rec[k]] >= 0 || v.prim_at[w.tri_rec[k]] >= 0; v.prim_at[w.tri_rec[k]] = (int32_t)(r.tri_order[k] + n_spheres); }
    for (size_t k = 0; k < w.sph_rec.size(); ++k) { clash += v.node_at[w.sph_rec[k]] >= 0 || v.prim_at[w.sph_rec[k]] >= 0; v.prim_at[w.sph_rec[k]] = (int32_t)r.sph_order[k]; }
    CHECK(clash == 0, "%s: %zu records assigned twice", what, clash);
    CHECK(w.node_rec[0] == 0, "%s: the root is not record 0", what);
    std::vector<int> seen(boxes.size(), 0);
    uint32_t max_level = 0;
    size_t n_nodes = 0;
    walk8(v, boxes, 0, 1, seen, max_level, n_nodes);
    size_t bad = 0;
    for (int s : seen) bad += s != 1;
    CHECK(bad == 0, "%s: %zu primitives not under exactly one BVH8 leaf slot", what, bad);
    CHECK(n_nodes == w.nodes.size() && max_level == w.depth, "%s: BVH8 walk saw %zu nodes / %u levels, builder says %zu / %u", what, n_nodes, max_level, w.nodes.size(), w.depth);
    // a BVH2 leaf's primitives stay in a row (the BVH2 arrays address them as first, first+1, ...)
    for (const Node64& nd : r.nodes)
        for (uint32_t ref : {nd.left, nd.right}) {
            if (ref == REF_EMPTY || !(ref & REF_LEAF)) continue;
            const std::vector<uint32_t>& map = (ref & REF_SPHERE) ? w.sph_rec : w.tri_rec;
            const uint32_t first = ref & REF_FIRST_MASK, cnt = ((ref >> 28) & 3u) + 1u;
            for (uint32_t j = 1; j < cnt; ++j) CHECK(map[first + j] == map[first] + j, "%s: leaf primitives not in a row", what);
        }
    printf("%-28s BVH8: %zu nodes, %u levels, %.2f children per node\n", what, w.nodes.size(), w.depth,
           w.nodes.empty() ? 0.0 : (double)(w.nodes.size() - 1 + r.nodes.size() + 1) / (double)w.nodes.size());
}

static size_t last_nodes = 0;
static uint32_t last_depth = 0;
static std::vector<uint32_t> last_tri_order, last_sph_order;
static std::vector<Node4> last_n4;
static void check_tree(const char* what, const std::vector<PrimBox>& boxes, size_t n_spheres) {
    BvhResult r;
    build_bvh(boxes, n_spheres, r);
    last_nodes = r.nodes.size();
    last_depth = r.depth;
    last_tri_order = r.tri_order;
    last_sph_order = r.sph_order;
    last_n4.clear();
    std::vector<int> seen(boxes.size(), 0);
    uint32_t max_depth = 0;
    if (!boxes.empty()) walk(r, boxes, n_spheres, 0, 0, seen, max_depth);
    size_t bad = 0;
    for (int s : seen) bad += s != 1;
    CHECK(bad == 0, "%s: %zu primitives not in exactly one leaf", what, bad);
    CHECK(max_depth <= (uint32_t)SAH_MAX_DEPTH && r.depth <= (uint32_t)SAH_MAX_DEPTH && max_depth == r.depth, "%s: depth %u (recorded %u)", what, max_depth, r.depth);
    CHECK(r.tri_order.size() == boxes.size() - n_spheres && r.sph_order.size() == n_spheres, "%s: order sizes", what);
    // derived node formats stay conservative
    std::vector<Node32> n32;
    narrow_nodes(r.nodes, n32);
    for (size_t i = 0; i < r.nodes.size(); ++i)
        for (int a = 0; a < 3; ++a) {
            CHECK((double)n32[i].lo_l[a] <= r.nodes[i].lo_l[a] && (double)n32[i].hi_l[a] >= r.nodes[i].hi_l[a], "%s: f32 left box", what);
            if (r.nodes[i].right != REF_EMPTY)
                CHECK((double)n32[i].lo_r[a] <= r.nodes[i].lo_r[a] && (double)n32[i].hi_r[a] >= r.nodes[i].hi_r[a], "%s: f32 right box", what);
        }
    std::vector<Node16> n16;
    QuantGrid g;
    if (!n32.empty() && quantize_nodes(n32, n16, g, 1e300)) {
        for (size_t i = 0; i < n32.size(); ++i)
            for (int a = 0; a < 3; ++a) {
                const long double lo = (long double)g.origin[a] + (long double)n16[i].lo_l[a] * g.cell[a];
                const long double hi = (long double)g.origin[a] + (long double)n16[i].hi_l[a] * g.cell[a];
                CHECK(lo <= n32[i].lo_l[a] && hi >= n32[i].hi_l[a], "%s: u16 left box of node %zu axis %d", what, i, a);
            }
        std::vector<Node4> n4;
        uint32_t need = 0;
        CHECK(widen_nodes(r.nodes, g, n4, need), "%s: widen_nodes failed", what);
        size_t leaves = 0;
        for (const Node4& nd : n4)
            for (int k = 0; k < 4; ++k) {
                if (nd.child[k] == REF_EMPTY) continue;
                if (nd.child[k] & REF_LEAF) leaves += ((nd.child[k] >> 28) & 3u) + 1u; else CHECK(nd.child[k] < n4.size(), "%s: BVH4 child index", what);
            }
        CHECK(leaves == boxes.size(), "%s: BVH4 holds %zu of %zu primitives", what, leaves, boxes.size());
        CHECK(need <= 3u * (r.depth + 1), "%s: BVH4 stack need %u", what, need);
        last_n4 = n4;
    }
    std::vector<Node4f> n4f;
    uint32_t need_f = 0;
    if (!r.nodes.empty()) CHECK(widen_nodes_f32(r.nodes, n4f, need_f), "%s: widen_nodes_f32 failed", what);
    check_wide8(what, r, boxes, n_spheres);
    printf("%-28s n=%8zu nodes=%8zu depth=%2u max_leaf=%u\n", what, boxes.size(), r.nodes.size(), r.depth, r.max_leaf);
}

int main() {
    std::mt19937_64 rng(42);
    std::uniform_real_distribution<double> U(-100.0, 100.0), S(0.01, 2.0);
    auto random_boxes = [&](size_t n) {
        std::vector<PrimBox> b(n);
        for (auto& p : b)
            for (int a = 0; a < 3; ++a) { const double c = U(rng), e = S(rng); p.lo[a] = c - e; p.hi[a] = c + e; }
        return b;
    };
    check_tree("empty", {}, 0);
    check_tree("one sphere", random_boxes(1), 1);
    check_tree("three triangles", random_boxes(3), 0);
    check_tree("mixed small", random_boxes(37), 11);
    check_tree("mixed 50k (sequential)", random_boxes(50000), 7000);
    {   // far from the origin with tiny extents, flat boxes (a ground plane's triangles), a huge box next to small ones:
        // the node-local grids must still enclose (exponent clamps, origin rounding)
        std::vector<PrimBox> b = random_boxes(3000);
        for (size_t i = 0; i < b.size(); ++i)
            for (int a = 0; a < 3; ++a) {
                if (i % 3 == 0) { b[i].lo[a] = 1.0e6 + 1e-4 * b[i].lo[a]; b[i].hi[a] = 1.0e6 + 1e-4 * b[i].hi[a]; }
                if (i % 5 == 0 && a == 2) b[i].hi[a] = b[i].lo[a];
            }
        for (int a = 0; a < 3; ++a) { b[7].lo[a] = -1.0e4; b[7].hi[a] = 1.0e4; }
        b[8].lo[0] = b[8].hi[0] = 0.0;
        check_tree("awkward boxes", b, 500);
    }
    {   // coincident primitives: every SAH split is degenerate
        std::vector<PrimBox> b(5000, random_boxes(1)[0]);
        check_tree("5000 coincident", b, 100);
    }
    {   // two-level build: clusters of boxes as groups, the rest loose; the stitched tree is an ordinary one
        std::vector<PrimBox> b = random_boxes(60000);
        const size_t ns = 300;
        std::vector<uint32_t> group_of(b.size(), 0);
        uint32_t n_groups = 0;
        for (size_t i = ns + 5000; i + 700 <= b.size(); i += 700) {
            ++n_groups;
            const size_t m = n_groups % 7 == 0 ? 3 : 40 + (n_groups * 37) % 600;          // a few tiny groups among them
            double c[3] = {U(rng), U(rng), U(rng)};
            for (size_t k = 0; k < m; ++k) {
                group_of[i + k] = n_groups;
                for (int a = 0; a < 3; ++a) { const double e = 0.05 * S(rng), x = c[a] + 0.02 * U(rng); b[i + k].lo[a] = x - e; b[i + k].hi[a] = x + e; }
            }
        }
        BvhResult r;
        build_bvh_grouped(b, ns, group_of, n_groups, r);
        std::vector<int> seen(b.size(), 0);
        uint32_t max_depth = 0;
        walk(r, b, ns, 0, 0, seen, max_depth);
        size_t bad = 0;
        for (int s_ : seen) bad += s_ != 1;
        CHECK(bad == 0, "grouped build: %zu primitives not in exactly one leaf", bad);
        CHECK(max_depth == r.depth && r.depth <= (uint32_t)BVH_MAX_DEPTH, "grouped build: depth %u (recorded %u)", max_depth, r.depth);
        CHECK(r.tri_order.size() == b.size() - ns && r.sph_order.size() == ns, "grouped build: order sizes");
        check_wide8("grouped 60k", r, b, ns);
        printf("%-28s n=%8zu nodes=%8zu depth=%2u groups=%u\n", "grouped 60k", b.size(), r.nodes.size(), r.depth, n_groups);
        BvhResult r1;
        setenv("RT_HIP_BUILD_THREADS", "1", 1);
        build_bvh_grouped(b, ns, group_of, n_groups, r1);
        unsetenv("RT_HIP_BUILD_THREADS");
        CHECK(r1.tri_order == r.tri_order && r1.nodes.size() == r.nodes.size() &&
              memcmp(r1.nodes.data(), r.nodes.data(), r.nodes.size() * sizeof(Node64)) == 0, "grouped build depends on the thread count");
    }
    const std::vector<PrimBox> big = random_boxes(300000);
    setenv("RT_HIP_BUILD_THREADS", "4", 1);
    check_tree("mixed 300k (4 threads)", big, 20000);
    const size_t nodes4 = last_nodes;
    const uint32_t depth4 = last_depth;
    const std::vector<uint32_t> tri4 = last_tri_order, sph4 = last_sph_order;
    const std::vector<Node4> wide4 = last_n4;
    setenv("RT_HIP_BUILD_THREADS", "1", 1);
    check_tree("mixed 300k (1 thread)", big, 20000);
    CHECK(nodes4 == last_nodes && depth4 == last_depth, "threaded build differs from the sequential one: %zu/%u vs %zu/%u", nodes4, depth4, last_nodes, last_depth);
    // the same tree whatever the thread count: the primitives sit in the leaves in the same order, and the BVH4 (numbered
    // in pre-order from the topology alone) is the same array byte for byte
    CHECK(tri4 == last_tri_order && sph4 == last_sph_order, "leaf order depends on the thread count");
    CHECK(!wide4.empty() && wide4.size() == last_n4.size() && memcmp(wide4.data(), last_n4.data(), wide4.size() * sizeof(Node4)) == 0,
          "BVH4 depends on the thread count (%zu vs %zu nodes)", wide4.size(), last_n4.size());
    setenv("RT_HIP_BUILD_THREADS", "7", 1);
    check_tree("mixed 300k (7 threads)", big, 20000);
    CHECK(tri4 == last_tri_order && wide4.size() == last_n4.size() && memcmp(wide4.data(), last_n4.data(), wide4.size() * sizeof(Node4)) == 0,
          "7-thread build differs from the 4-thread one");
    printf(fails ? "FAILED (%d)\n" : "OK\n", fails);
    return fails ? 1 : 0;
}
