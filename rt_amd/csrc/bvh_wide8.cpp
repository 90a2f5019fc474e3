// bvh_wide8.cpp — BVH2 -> BVH8 with 8-bit node-local boxes (rt_layout.h: Node8; bvh_build.h: widen_nodes8).
#include <algorithm>
#include <cmath>
#include <cstring>
#include <limits>

#include "bvh_build.h"

namespace rt {
namespace {

struct Ent {
    uint32_t ref;
    double lo[3], hi[3];
};
inline double area(const Ent& e) {
    const double x = e.hi[0] - e.lo[0], y = e.hi[1] - e.lo[1], z = e.hi[2] - e.lo[2];
    return x * y + y * z + z * x;
}
inline void children2(const Node64& n, Ent* e, int& c) {
    c = 0;
    e[c].ref = n.left;
    for (int k = 0; k < 3; ++k) { e[c].lo[k] = n.lo_l[k]; e[c].hi[k] = n.hi_l[k]; }
    ++c;
    if (n.right != REF_EMPTY) {
        e[c].ref = n.right;
        for (int k = 0; k < 3; ++k) { e[c].lo[k] = n.lo_r[k]; e[c].hi[k] = n.hi_r[k]; }
        ++c;
    }
}
inline float f32_down(double x) {
    float f = (float)x;
    if ((double)f > x) f = std::nextafterf(f, -std::numeric_limits<float>::infinity());
    return f;
}

struct Builder8 {
    const std::vector<Node64>& in;
    Wide8& out;
    bool ok = true;
    uint32_t depth = 0;

    // the children of the BVH8 node that replaces BVH2 node n2: its own two, then the internal child with the largest
    // box is replaced by its two children, until there are eight (or only leaves are left)
    int collapse(uint32_t n2, Ent* e) const {
        int c = 0;
        children2(in[n2], e, c);
        while (c < 8) {
            int best = -1;
            double best_area = -1.0;
            for (int i = 0; i < c; ++i)
                if (!(e[i].ref & REF_LEAF) && area(e[i]) > best_area) { best = i; best_area = area(e[i]); }
            if (best < 0) break;
            Ent sub[2];
            int cs = 0;
            children2(in[e[best].ref], sub, cs);
            e[best] = sub[0];
            if (cs == 2) e[c++] = sub[1];
        }
        return c;
    }

    // records of the subtree below n2, its own node record not counted
    size_t count(uint32_t n2) const {
        Ent e[9];
        const int c = collapse(n2, e);
        size_t k = 0;
        for (int i = 0; i < c; ++i) {
            if (e[i].ref & REF_LEAF) k += ((e[i].ref >> 28) & 3u) + 1u;
            else k += 1 + count(e[i].ref);
        }
        return k;
    }

    // Octant order (Ylitie et al., section 3.1): child i in slot s costs -(c_i - c_node) . (+-1, +-1, +-1) with the signs
    // given by the bits of s (bit k set = the + side of axis k), so that for a ray running in + direction on every axis
    // ascending slots are front to back, and for any other octant ascending (slot XOR octant).  Greedy assignment: the
    // pair with the largest gain first.
    static void assign_slots(const Ent* e, int c, const double* nlo, const double* nhi, int* slot_of) {
        double d[8][3];
        for (int i = 0; i < c; ++i)
            for (int k = 0; k < 3; ++k) d[i][k] = 0.5 * (e[i].lo[k] + e[i].hi[k]) - 0.5 * (nlo[k] + nhi[k]);
        bool child_done[8] = {}, slot_used[8] = {};
        for (int i = 0; i < c; ++i) slot_of[i] = -1;
        for (int round = 0; round < c; ++round) {
            int bi = -1, bs = -1;
            double best = -std::numeric_limits<double>::infinity();
            for (int i = 0; i < c; ++i) {
                if (child_done[i]) continue;
                for (int s = 0; s < 8; ++s) {
                    if (slot_used[s]) continue;
                    double g = 0.0;
                    for (int k = 0; k < 3; ++k) g += ((s >> k) & 1) ? d[i][k] : -d[i][k];
                    if (g > best) { best = g; bi = i; bs = s; }
                }
            }
            child_done[bi] = true;
            slot_used[bs] = true;
            slot_of[bi] = bs;
        }
    }

    // the real number a plane byte stands for: origin + q * 2^(ex - 127); exact in a 64-bit mantissa (make_grid keeps
    // the cell within 2^45 of the origin's magnitude)
    static long double plane(float origin, int ex, int q) { return (long double)origin + std::ldexp((long double)q, ex - 127); }

    // grid of one axis: origin = lo rounded down to f32, ex = the smallest biased exponent with origin + 255 * 2^(ex-127) >= hi
    bool make_grid(double lo, double hi, float& origin, int& ex) {
        origin = f32_down(lo);
        if (!std::isfinite(origin) || !std::isfinite(hi)) return false;
        const long double ext = (long double)hi - (long double)origin;
        int e = -100;
        if (ext > 0.0L) {
            int x;
            std::frexp((double)(ext / 255.0L), &x);          // ext / 255 = m * 2^x, m in [0.5, 1)  =>  2^x > ext / 255
            e = std::max(e, x - 1);
            while (std::ldexp(255.0L, e) < ext) ++e;
        }
        if (origin != 0.0f) {                                // (a cell far below the origin's last bit buys nothing)
            int xo;
            std::frexp((double)origin, &xo);
            e = std::max(e, xo - 45);
        }
        if (e > 100) return false;
        ex = e + 127;
        return true;
    }

    // Writes node `index` (its record: rec) for BVH2 node n2 and lays its block out at `cursor`: internal children,
    // leaf primitives, then the blocks of the internal children one after the other.
    void emit(uint32_t n2, uint32_t index, uint32_t rec, uint32_t& next_index, uint32_t& cursor, uint32_t level) {
        Ent e[9];
        const int c = collapse(n2, e);
        depth = std::max(depth, level);
        double nlo[3], nhi[3];
        for (int k = 0; k < 3; ++k) {
            nlo[k] = e[0].lo[k]; nhi[k] = e[0].hi[k];
            for (int i = 1; i < c; ++i) { nlo[k] = std::min(nlo[k], e[i].lo[k]); nhi[k] = std::max(nhi[k], e[i].hi[k]); }
        }
        int slot_of[8];
        assign_slots(e, c, nlo, nhi, slot_of);
        int child_in_slot[8];
        for (int s = 0; s < 8; ++s) child_in_slot[s] = -1;
        for (int i = 0; i < c; ++i) child_in_slot[slot_of[i]] = i;

        Node8 nd;
        std::memset(&nd, 0, sizeof(nd));
        int ex[3] = {27, 27, 27};
        for (int k = 0; k < 3; ++k) {
            if (!make_grid(nlo[k], nhi[k], nd.origin[k], ex[k])) { ok = false; ex[k] = 127; }
            nd.ex[k] = (uint8_t)ex[k];
        }
        for (int s = 0; s < 8; ++s)
            for (int k = 0; k < 3; ++k) { nd.qlo[k][s] = 255; nd.qhi[k][s] = 0; }      // an empty slot's box is inverted
        // the block: internal children first, then the primitives of the leaf children, both in slot order
        uint32_t n_int = 0, n_prim = 0;
        for (int s = 0; s < 8; ++s) {
            const int i = child_in_slot[s];
            if (i < 0) continue;
            if (e[i].ref & REF_LEAF) n_prim += ((e[i].ref >> 28) & 3u) + 1u; else ++n_int;
        }
        nd.base = cursor;
        const uint32_t first_child_rec = cursor, first_child_index = next_index;
        uint32_t prim_rec = cursor + n_int;
        cursor += n_int + n_prim;
        next_index += n_int;
        uint32_t rank = 0;
        uint32_t child_n2[8], child_idx[8], child_rec[8];
        for (int s = 0; s < 8; ++s) {
            const int i = child_in_slot[s];
            if (i < 0) continue;
            for (int k = 0; k < 3; ++k) {
                const long double cell = std::ldexp(1.0L, ex[k] - 127);
                long ql = (long)std::floor(((long double)e[i].lo[k] - (long double)nd.origin[k]) / cell);
                long qh = (long)std::ceil(((long double)e[i].hi[k] - (long double)nd.origin[k]) / cell);
                ql = std::min(std::max(ql, 0L), 255L);
                qh = std::min(std::max(qh, 0L), 255L);
                while (ql > 0 && plane(nd.origin[k], ex[k], (int)ql) > (long double)e[i].lo[k]) --ql;
                while (qh < 255 && plane(nd.origin[k], ex[k], (int)qh) < (long double)e[i].hi[k]) ++qh;
                if (plane(nd.origin[k], ex[k], (int)ql) > (long double)e[i].lo[k] || plane(nd.origin[k], ex[k], (int)qh) < (long double)e[i].hi[k]) ok = false;
                nd.qlo[k][s] = (uint8_t)ql;
                nd.qhi[k][s] = (uint8_t)qh;
            }
            if (e[i].ref & REF_LEAF) {
                const uint32_t cnt = ((e[i].ref >> 28) & 3u) + 1u, first = e[i].ref & REF_FIRST_MASK;
                nd.pmask |= ((1u << cnt) - 1u) << (4 * s);
                std::vector<uint32_t>& map = (e[i].ref & REF_SPHERE) ? out.sph_rec : out.tri_rec;
                for (uint32_t j = 0; j < cnt; ++j) {
                    if (first + j >= map.size()) { ok = false; break; }
                    map[first + j] = prim_rec + j;
                }
                prim_rec += cnt;
            } else {
                nd.inib |= 0xFu << (4 * s);
                child_n2[rank] = e[i].ref;
                child_idx[rank] = first_child_index + rank;
                child_rec[rank] = first_child_rec + rank;
                ++rank;
            }
        }
        out.nodes[index] = nd;
        out.node_rec[index] = rec;
        for (uint32_t r = 0; r < rank; ++r) emit(child_n2[r], child_idx[r], child_rec[r], next_index, cursor, level + 1);
    }
};

// number of BVH8 nodes of the subtree of n2 (itself included)
size_t count_nodes(const Builder8& b, uint32_t n2) {
    Ent e[9];
    const int c = b.collapse(n2, e);
    size_t k = 1;
    for (int i = 0; i < c; ++i)
        if (!(e[i].ref & REF_LEAF)) k += count_nodes(b, e[i].ref);
    return k;
}

}  // namespace

bool widen_nodes8(const std::vector<Node64>& in, size_t n_triangles, size_t n_spheres, Wide8& out) {
    out = Wide8();
    if (in.empty()) return false;
    Builder8 b{in, out};
    const size_t n_nodes = count_nodes(b, 0);
    const size_t n_records = n_nodes + n_triangles + n_spheres;
    if (n_records + 16 >= (size_t)REF_FIRST_MASK) return false;       // BVH2 leaf references count pool records in 28 bits
    out.nodes.resize(n_nodes);
    out.node_rec.resize(n_nodes);
    out.tri_rec.assign(n_triangles, 0xFFFFFFFFu);
    out.sph_rec.assign(n_spheres, 0xFFFFFFFFu);
    uint32_t next_index = 1, cursor = 1;
    b.emit(0, 0, 0, next_index, cursor, 1);
    if (!b.ok || next_index != n_nodes || cursor != n_records || b.depth > (uint32_t)BVH8_MAX_DEPTH) return false;
    for (uint32_t r : out.tri_rec) if (r == 0xFFFFFFFFu) return false;
    for (uint32_t r : out.sph_rec) if (r == 0xFFFFFFFFu) return false;
    out.n_records = n_records;
    out.depth = b.depth;
    return true;
}

}  // namespace rt
