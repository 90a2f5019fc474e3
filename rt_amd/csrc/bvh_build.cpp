// bvh_build.cpp — binned-SAH BVH2 builder (host).  See bvh_build.h.
#include "bvh_build.h"

#include <algorithm>
#include <atomic>
#include <cmath>
#include <condition_variable>
#include <cstdlib>
#include <exception>
#include <functional>
#include <limits>
#include <mutex>
#include <system_error>
#include <thread>

namespace rt {
namespace {

#ifndef RT_NBINS
#define RT_NBINS 16
#endif
// (32 bins, or an exact sweep over the sorted centroids for ranges of up to 64 / 512 primitives, change the node visits per
// ray by less than 0.5 % on the 1 M and the 5 M triangle scenes: A/B of round 3)
constexpr int NBINS = RT_NBINS;
constexpr double COST_TRAVERSE = 1.0;
#ifndef RT_COST_PRIM
#define RT_COST_PRIM 1.0
#endif
#ifndef RT_LEAF_MAX
#define RT_LEAF_MAX MAX_LEAF
#endif
constexpr double COST_PRIM = RT_COST_PRIM;
constexpr double INF = std::numeric_limits<double>::infinity();

struct Box {
    double lo[3] = {INF, INF, INF}, hi[3] = {-INF, -INF, -INF};
    void grow(const double* l, const double* h) {
        for (int k = 0; k < 3; ++k) {
            lo[k] = std::min(lo[k], l[k]);
            hi[k] = std::max(hi[k], h[k]);
        }
    }
    void grow(const Box& b) { grow(b.lo, b.hi); }
    void grow_point(const double* p) { grow(p, p); }
    double half_area() const {
        double ex = hi[0] - lo[0], ey = hi[1] - lo[1], ez = hi[2] - lo[2];
        return ex * ey + ey * ez + ez * ex;
    }
};

inline int ceil_log2(size_t n) {
    int l = 0;
    while ((size_t(1) << l) < n) ++l;
    return l;
}

// Subtrees of at most `task_threshold` primitives are built by worker threads: the top of the tree is built
// first and records them as tasks (placeholder references), the workers build each into its own BvhResult, and
// build_bvh stitches them in task order (deterministic, whatever the thread timing).
constexpr uint32_t TASK_REF = 0x70000000u;               // placeholder: TASK_REF | task index (node indices stay below)
struct Task {
    size_t first = 0, count = 0;
    int depth = 0;
    uint32_t parent = 0;
    int side = 0;                                         // 0: parent's left, 1: right
    Box bounds;
    uint32_t ref = REF_EMPTY;
    BvhResult result;
};

// A few threads that stay up for the length of one build: run(n, fn) calls fn(k) once for every k in [0, n) on the
// team and on the caller, and returns when all calls have returned.  What a run computes never depends on which thread
// took which k (callers merge per-k results in k order, or the merge is order-free: min, max, integer sums).
class Team {
public:
    explicit Team(unsigned threads) {
        // a thread that cannot be started is not an error: the team is then smaller (down to the caller alone)
        try {
            for (unsigned i = 1; i < threads; ++i) workers.emplace_back([this] { loop(); });
        } catch (const std::system_error&) {
        }
    }
    ~Team() {
        {
            std::lock_guard<std::mutex> l(m);
            stop = true;
            ++generation;
        }
        cv_job.notify_all();
        for (auto& t : workers) t.join();
    }
    unsigned size() const { return (unsigned)workers.size() + 1u; }
    void run(size_t n, const std::function<void(size_t)>& fn) {
        if (workers.empty() || n <= 1) {
            for (size_t k = 0; k < n; ++k) fn(k);      // (on the caller: exceptions propagate as they are)
            return;
        }
        {
            std::lock_guard<std::mutex> l(m);
            job = &fn;
            total = n;
            next.store(0);
            busy = workers.size();
            ++generation;
        }
        cv_job.notify_all();
        drain(fn, n);
        std::exception_ptr err;
        {
            std::unique_lock<std::mutex> l(m);
            cv_done.wait(l, [&] { return busy == 0; });
            job = nullptr;
            err = failure;
            failure = nullptr;
        }
        // an exception inside fn (bad_alloc from a vector, say) on whichever thread is raised again HERE, on the caller:
        // a worker thread has no handler, and the entry points of the library catch on their own thread
        if (err) std::rethrow_exception(err);
    }

private:
    void drain(const std::function<void(size_t)>& fn, size_t n) {
        for (;;) {
            const size_t k = next.fetch_add(1);
            if (k >= n) return;
            try {
                fn(k);
            } catch (...) {
                std::lock_guard<std::mutex> l(m);
                if (!failure) failure = std::current_exception();
                next.store(n);             // nobody starts another piece
            }
        }
    }
    void loop() {
        unsigned long long seen = 0;
        for (;;) {
            const std::function<void(size_t)>* fn;
            size_t n;
            {
                std::unique_lock<std::mutex> l(m);
                cv_job.wait(l, [&] { return generation != seen; });
                seen = generation;
                if (stop) return;
                fn = job;
                n = total;
            }
            drain(*fn, n);
            {
                std::lock_guard<std::mutex> l(m);
                if (--busy == 0) cv_done.notify_one();
            }
        }
    }
    std::vector<std::thread> workers;
    std::mutex m;
    std::condition_variable cv_job, cv_done;
    const std::function<void(size_t)>* job = nullptr;
    size_t total = 0, busy = 0;
    std::atomic<size_t> next{0};
    unsigned long long generation = 0;
    bool stop = false;
    std::exception_ptr failure;        // first exception of the run in flight (under m)
};

// [0, n) in pieces of `piece` elements: piece k is [k*piece, min(n, (k+1)*piece))
inline size_t pieces_of(size_t n, size_t piece) { return (n + piece - 1) / piece; }

// Ranges of at least this many primitives are swept and partitioned in pieces (at the top of the tree by the team,
// inside a worker's subtree one piece after the other): the partition of such a range is the STABLE one whoever runs
// it, so the tree and the order of the primitives in its leaves do not depend on the number of threads.
constexpr size_t PIECEWISE_MIN = 65536, PIECE = 16384;

struct Shared {
    const std::vector<PrimBox>& pb;
    size_t nsph;
    std::vector<uint32_t> idx;
    std::vector<double> cent;      // 3 per prim
    std::vector<uint32_t> spare;   // the stable partition's second buffer (a builder uses the range it owns in idx)
    size_t first_solo = SIZE_MAX;  // items from this index on never share a leaf (build_bvh_grouped: the groups' boxes)
    Shared(const std::vector<PrimBox>& b, size_t ns, Team& team) : pb(b), nsph(ns) {
        idx.resize(b.size());
        cent.resize(3 * b.size());
        if (b.size() >= PIECEWISE_MIN) spare.resize(b.size());
        team.run(pieces_of(b.size(), PIECE), [&](size_t k) {
            for (size_t i = k * PIECE; i < std::min(b.size(), (k + 1) * PIECE); ++i) {
                idx[i] = (uint32_t)i;
                for (int c = 0; c < 3; ++c) cent[3 * i + c] = 0.5 * (b[i].lo[c] + b[i].hi[c]);
            }
        });
    }
};

struct Builder {
    const std::vector<PrimBox>& pb;
    size_t nsph;
    std::vector<uint32_t>& idx;     // shared; a builder only touches its own range
    const std::vector<double>& cent;
    BvhResult& out;
    std::vector<Task>* tasks = nullptr;
    size_t task_threshold = 0;
    std::vector<uint32_t>& spare;   // shared like idx: the stable partition's second buffer, same ranges
    Team* team = nullptr;           // set for the builder of the top of the tree (the only one running at that time)
    size_t first_solo;

    Builder(Shared& sh, BvhResult& o)
        : pb(sh.pb), nsph(sh.nsph), idx(sh.idx), cent(sh.cent), out(o), spare(sh.spare), first_solo(sh.first_solo) {}

    void each_piece(size_t n_pieces, const std::function<void(size_t)>& fn) const {
        if (team) team->run(n_pieces, fn);
        else for (size_t k = 0; k < n_pieces; ++k) fn(k);
    }

    bool homogeneous(size_t first, size_t count) const {
        bool s0 = idx[first] < nsph;
        if (count > 1 && idx[first] >= first_solo) return false;
        for (size_t i = first + 1; i < first + count; ++i)
            if ((idx[i] < nsph) != s0 || idx[i] >= first_solo) return false;
        return true;
    }

    uint32_t make_leaf(size_t first, size_t count, int depth) {
        bool sph = idx[first] < nsph;
        uint32_t start;
        if (sph) {
            start = (uint32_t)out.sph_order.size();
            for (size_t i = first; i < first + count; ++i) out.sph_order.push_back(idx[i]);
        } else {
            start = (uint32_t)out.tri_order.size();
            for (size_t i = first; i < first + count; ++i) out.tri_order.push_back(idx[i] - (uint32_t)nsph);
        }
        out.depth = std::max(out.depth, (uint32_t)depth);
        out.max_leaf = std::max(out.max_leaf, (uint32_t)count);
        return REF_LEAF | (sph ? REF_SPHERE : 0u) | ((uint32_t)(count - 1) << 28) | start;
    }

    // Builds the subtree over idx[first, first+count); returns its child reference (which sits at
    // `depth`) and its bounds.  Invariant: ceil_log2(count) <= SAH_MAX_DEPTH - depth.
    uint32_t build(size_t first, size_t count, int depth, Box& bounds) {
        if (tasks && depth > 0 && count <= task_threshold) {
            Task t;
            t.first = first; t.count = count; t.depth = depth;
            tasks->push_back(std::move(t));
            return TASK_REF | (uint32_t)(tasks->size() - 1);
        }
        Box cb;
        bounds = Box();
        const bool piecewise = count >= PIECEWISE_MIN;
        const size_t n_pieces = piecewise ? pieces_of(count, PIECE) : 1;
        auto piece_end = [&](size_t k) { return first + std::min(count, (k + 1) * PIECE); };
        if (piecewise) {
            std::vector<Box> part(2 * n_pieces);
            each_piece(n_pieces, [&](size_t k) {
                Box b, c;
                for (size_t i = first + k * PIECE; i < piece_end(k); ++i) {
                    const PrimBox& p = pb[idx[i]];
                    b.grow(p.lo, p.hi);
                    c.grow_point(&cent[3 * idx[i]]);
                }
                part[2 * k] = b;
                part[2 * k + 1] = c;
            });
            for (size_t k = 0; k < n_pieces; ++k) { bounds.grow(part[2 * k]); cb.grow(part[2 * k + 1]); }
        } else {
            for (size_t i = first; i < first + count; ++i) {
                const PrimBox& p = pb[idx[i]];
                bounds.grow(p.lo, p.hi);
                cb.grow_point(&cent[3 * idx[i]]);
            }
        }
        const bool can_leaf = count <= (size_t)RT_LEAF_MAX && homogeneous(first, count);
        if (count == 1) return make_leaf(first, count, depth);

        // --- binned SAH over the three axes
        double best_cost = INF;
        int best_axis = -1, best_split = -1;
        // (a large range: the bins of the three axes are filled piece by piece and merged; counts are integers and boxes
        // are min/max, so the merged bins equal the ones a single sweep fills)
        struct Bins { size_t cnt[3][NBINS]; Box bb[3][NBINS]; };
        std::vector<Bins> piece_bins;
        if (piecewise) {
            piece_bins.resize(n_pieces);
            each_piece(n_pieces, [&](size_t k) {
                Bins& B = piece_bins[k];
                for (int ax = 0; ax < 3; ++ax) {
                    for (int b = 0; b < NBINS; ++b) B.cnt[ax][b] = 0;
                    const double c0 = cb.lo[ax], ext = cb.hi[ax] - cb.lo[ax];
                    if (!(ext > 0.0)) continue;
                    const double scale = NBINS / ext;
                    for (size_t i = first + k * PIECE; i < piece_end(k); ++i) {
                        int b = (int)((cent[3 * idx[i] + ax] - c0) * scale);
                        b = std::min(std::max(b, 0), NBINS - 1);
                        B.cnt[ax][b]++;
                        B.bb[ax][b].grow(pb[idx[i]].lo, pb[idx[i]].hi);
                    }
                }
            });
        }
        for (int ax = 0; ax < 3; ++ax) {
            double c0 = cb.lo[ax], ext = cb.hi[ax] - cb.lo[ax];
            if (!(ext > 0.0)) continue;
            size_t cnt[NBINS] = {0};
            Box bb[NBINS];
            double scale = NBINS / ext;
            if (piecewise) {
                for (size_t k = 0; k < n_pieces; ++k)
                    for (int b = 0; b < NBINS; ++b) {
                        cnt[b] += piece_bins[k].cnt[ax][b];
                        if (piece_bins[k].cnt[ax][b]) bb[b].grow(piece_bins[k].bb[ax][b]);
                    }
            } else {
                for (size_t i = first; i < first + count; ++i) {
                    int k = (int)((cent[3 * idx[i] + ax] - c0) * scale);
                    k = std::min(std::max(k, 0), NBINS - 1);
                    cnt[k]++;
                    bb[k].grow(pb[idx[i]].lo, pb[idx[i]].hi);
                }
            }
            double right_area[NBINS];
            size_t right_cnt[NBINS];
            Box acc;
            size_t c = 0;
            for (int k = NBINS - 1; k >= 1; --k) {
                if (cnt[k]) acc.grow(bb[k]);
                c += cnt[k];
                right_area[k] = c ? acc.half_area() : 0.0;
                right_cnt[k] = c;
            }
            acc = Box();
            c = 0;
            for (int s = 1; s < NBINS; ++s) {
                if (cnt[s - 1]) acc.grow(bb[s - 1]);
                c += cnt[s - 1];
                if (!c || !right_cnt[s]) continue;
                double cost = acc.half_area() * (double)c + right_area[s] * (double)right_cnt[s];
                if (cost < best_cost) { best_cost = cost; best_axis = ax; best_split = s; }
            }
        }
        const double area = bounds.half_area();
        if (can_leaf) {
            double split_cost = best_axis >= 0 && area > 0.0 ? COST_TRAVERSE + COST_PRIM * best_cost / area : INF;
            if ((double)count * COST_PRIM <= split_cost) return make_leaf(first, count, depth);
        }

        // --- partition
        const int remaining = SAH_MAX_DEPTH - depth;   // levels available below this reference
        size_t mid = first;
        if (best_axis >= 0) {
            double c0 = cb.lo[best_axis], scale = NBINS / (cb.hi[best_axis] - cb.lo[best_axis]);
            auto goes_left = [&](uint32_t p) {
                int k = (int)((cent[3 * p + best_axis] - c0) * scale);
                k = std::min(std::max(k, 0), NBINS - 1);
                return k < best_split;
            };
            if (piecewise) {
                // stable partition through the spare buffer: count per piece, place every piece's left and right run
                // behind those of the pieces before it, copy back
                std::vector<size_t> n_left(n_pieces + 1, 0);
                each_piece(n_pieces, [&](size_t k) {
                    size_t c = 0;
                    for (size_t i = first + k * PIECE; i < piece_end(k); ++i) c += goes_left(idx[i]) ? 1 : 0;
                    n_left[k + 1] = c;
                });
                for (size_t k = 0; k < n_pieces; ++k) n_left[k + 1] += n_left[k];      // lefts before piece k+1
                const size_t all_left = n_left[n_pieces];
                each_piece(n_pieces, [&](size_t k) {
                    size_t l = first + n_left[k], r = first + all_left + (k * PIECE - n_left[k]);
                    for (size_t i = first + k * PIECE; i < piece_end(k); ++i) {
                        const uint32_t p = idx[i];
                        if (goes_left(p)) spare[l++] = p; else spare[r++] = p;
                    }
                });
                each_piece(n_pieces, [&](size_t k) {
                    std::copy(spare.begin() + first + k * PIECE, spare.begin() + piece_end(k), idx.begin() + first + k * PIECE);
                });
                mid = first + all_left;
            } else {
                auto it = std::partition(idx.begin() + first, idx.begin() + first + count, goes_left);
                mid = (size_t)(it - idx.begin());
            }
        }
        size_t nl = mid - first, nr = count - nl;
        if (nl == 0 || nr == 0 || ceil_log2(nl) > remaining - 1 || ceil_log2(nr) > remaining - 1) {
            // spatial median along the widest centroid axis: keeps the depth invariant
            int ax = 0;
            for (int k = 1; k < 3; ++k)
                if (cb.hi[k] - cb.lo[k] > cb.hi[ax] - cb.lo[ax]) ax = k;
            mid = first + count / 2;
            std::nth_element(idx.begin() + first, idx.begin() + mid, idx.begin() + first + count,
                             [&](uint32_t a, uint32_t b) { return cent[3 * a + ax] < cent[3 * b + ax]; });
            nl = mid - first;
            nr = count - nl;
        }

        uint32_t me = (uint32_t)out.nodes.size();
        out.nodes.emplace_back();
        Box bl, br;
        uint32_t l = build(first, nl, depth + 1, bl);
        uint32_t r = build(mid, nr, depth + 1, br);
        if (tasks) {
            if ((l & 0xF0000000u) == TASK_REF) { (*tasks)[l & 0x0FFFFFFFu].parent = me; (*tasks)[l & 0x0FFFFFFFu].side = 0; }
            if ((r & 0xF0000000u) == TASK_REF) { (*tasks)[r & 0x0FFFFFFFu].parent = me; (*tasks)[r & 0x0FFFFFFFu].side = 1; }
        }
        Node64& n = out.nodes[me];
        for (int k = 0; k < 3; ++k) {
            n.lo_l[k] = bl.lo[k]; n.hi_l[k] = bl.hi[k];
            n.lo_r[k] = br.lo[k]; n.hi_r[k] = br.hi[k];
        }
        n.left = l;
        n.right = r;
        for (uint32_t& p : n._pad) p = 0;
        out.depth = std::max(out.depth, (uint32_t)depth + 1);
        return me;
    }
};

inline float round_down(double x) {
    float f = (float)x;
    if ((double)f > x) f = std::nextafterf(f, -std::numeric_limits<float>::infinity());
    return f;
}
inline float round_up(double x) {
    float f = (float)x;
    if ((double)f < x) f = std::nextafterf(f, std::numeric_limits<float>::infinity());
    return f;
}

}  // namespace

static unsigned build_threads() {
    const char* e = getenv("RT_HIP_BUILD_THREADS");
    if (e && atoi(e) > 0) return (unsigned)atoi(e);
    const unsigned hw = std::thread::hardware_concurrency();
    return std::max(1u, std::min(hw ? hw : 1u, 16u));
}

void for_each_piece(size_t n_pieces, const std::function<void(size_t)>& fn) {
    Team crew(n_pieces > 1 ? build_threads() : 1u);
    crew.run(n_pieces, fn);
}

static void build_bvh_solo(const std::vector<PrimBox>& boxes, size_t n_spheres, BvhResult& out, size_t first_solo, bool serial);
void build_bvh(const std::vector<PrimBox>& boxes, size_t n_spheres, BvhResult& out) {
    build_bvh_solo(boxes, n_spheres, out, SIZE_MAX, false);
}
static void build_bvh_solo(const std::vector<PrimBox>& boxes, size_t n_spheres, BvhResult& out, size_t first_solo, bool serial) {
    out = BvhResult();
    const size_t n = boxes.size();
    if (n == 0) return;
    out.nodes.reserve(n / 2 + 16);
    out.tri_order.reserve(n - n_spheres);
    out.sph_order.reserve(n_spheres);
    const unsigned T = serial ? 1u : build_threads();
    const bool threaded = T > 1 && n >= 200000;
    Team crew(threaded ? T : 1u);
    Shared sh(boxes, n_spheres, crew);
    sh.first_solo = first_solo;
    Builder B(sh, out);
    Box bounds;
    // Probe: would the whole scene be a single leaf?  Then wrap it so that node 0 always exists.
    bool single = (n == 1 && first_solo == SIZE_MAX) || (n > 1 && n <= (size_t)RT_LEAF_MAX && B.homogeneous(0, n));
    if (single) {
        Box b;
        for (size_t i = 0; i < n; ++i) b.grow(boxes[i].lo, boxes[i].hi);
        out.nodes.emplace_back();
        uint32_t leaf = B.make_leaf(0, n, 1);
        Node64& nd = out.nodes[0];
        for (int k = 0; k < 3; ++k) {
            nd.lo_l[k] = b.lo[k]; nd.hi_l[k] = b.hi[k];
            nd.lo_r[k] = INF; nd.hi_r[k] = -INF;
        }
        nd.left = leaf;
        nd.right = REF_EMPTY;
        for (uint32_t& p : nd._pad) p = 0;
        return;
    }
    std::vector<Task> tasks;
    if (threaded) {
        B.tasks = &tasks;
        B.task_threshold = std::max<size_t>(4096, n / ((size_t)T * 8));
        B.team = &crew;             // the top of the tree: every large range is swept by the whole team
    }
    uint32_t root = B.build(0, n, 0, bounds);
    if (!tasks.empty()) {
        // the team takes the tasks from a shared counter; every task writes only its own Task and its own idx range
        crew.run(tasks.size(), [&](size_t k) {
            Task& t = tasks[k];
            t.result.nodes.reserve(t.count / 2 + 16);
            Builder b(sh, t.result);
            t.ref = b.build(t.first, t.count, t.depth, t.bounds);
        });
        // stitch in task order: where every task's nodes and leaf orders go follows from the sizes, then the team copies
        struct Base { size_t node, tri, sph; };
        std::vector<Base> base(tasks.size() + 1);
        base[0] = {out.nodes.size(), out.tri_order.size(), out.sph_order.size()};
        for (size_t k = 0; k < tasks.size(); ++k)
            base[k + 1] = {base[k].node + tasks[k].result.nodes.size(), base[k].tri + tasks[k].result.tri_order.size(),
                           base[k].sph + tasks[k].result.sph_order.size()};
        if (base.back().node >= (size_t)TASK_REF) { out = BvhResult(); return; }      // (cannot happen below 2^28 primitives)
        out.nodes.resize(base.back().node);
        out.tri_order.resize(base.back().tri);
        out.sph_order.resize(base.back().sph);
        crew.run(tasks.size(), [&](size_t k) {
            Task& t = tasks[k];
            const uint32_t node_base = (uint32_t)base[k].node, tri_base = (uint32_t)base[k].tri, sph_base = (uint32_t)base[k].sph;
            auto fix = [&](uint32_t ref) -> uint32_t {
                if (ref == REF_EMPTY) return ref;
                if (ref & REF_LEAF) return ref + ((ref & REF_SPHERE) ? sph_base : tri_base);     // `first` sits in the low bits
                return ref + node_base;
            };
            Node64* dst = out.nodes.data() + base[k].node;
            for (size_t i = 0; i < t.result.nodes.size(); ++i) {
                Node64 nd = t.result.nodes[i];
                nd.left = fix(nd.left);
                nd.right = fix(nd.right);
                dst[i] = nd;
            }
            std::copy(t.result.tri_order.begin(), t.result.tri_order.end(), out.tri_order.begin() + base[k].tri);
            std::copy(t.result.sph_order.begin(), t.result.sph_order.end(), out.sph_order.begin() + base[k].sph);
            t.ref = fix(t.ref);
            const uint32_t depth = t.result.depth, max_leaf = t.result.max_leaf;
            t.result = BvhResult();
            t.result.depth = depth;
            t.result.max_leaf = max_leaf;
        });
        for (Task& t : tasks) {
            out.depth = std::max(out.depth, t.result.depth);
            out.max_leaf = std::max(out.max_leaf, t.result.max_leaf);
            Node64& par = out.nodes[t.parent];
            double* lo = t.side ? par.lo_r : par.lo_l;
            double* hi = t.side ? par.hi_r : par.hi_l;
            for (int k = 0; k < 3; ++k) { lo[k] = t.bounds.lo[k]; hi[k] = t.bounds.hi[k]; }
            (t.side ? par.right : par.left) = t.ref;
        }
    }
    if (root & REF_LEAF) {
        // count in (1, MAX_LEAF] and SAH preferred a leaf: cannot happen (handled by `single`),
        // kept for safety.
        out.nodes.emplace_back();
        Node64& nd = out.nodes[0];
        for (int k = 0; k < 3; ++k) {
            nd.lo_l[k] = bounds.lo[k]; nd.hi_l[k] = bounds.hi[k];
            nd.lo_r[k] = INF; nd.hi_r[k] = -INF;
        }
        nd.left = root;
        nd.right = REF_EMPTY;
    }
}

// Two-level build.  group_of[p] > 0 puts primitive p (a triangle) into group group_of[p] (1 .. n_groups): every group
// gets a tree of its own, and the top of the final tree is built over the loose primitives and ONE box per group, so no
// split of the upper levels ever cuts through a group and no node mixes a group's triangles with anything else.  The result
// is an ordinary BvhResult: the groups' trees are stitched in where the top tree has their boxes.
void build_bvh_grouped(const std::vector<PrimBox>& boxes, size_t n_spheres, const std::vector<uint32_t>& group_of, uint32_t n_groups,
                       BvhResult& out) {
    const size_t n = boxes.size();
    if (!n_groups || group_of.size() != n) { build_bvh(boxes, n_spheres, out); return; }
    // ---- members of every group (original triangle indices, ascending) and the loose items
    std::vector<uint32_t> g_start(n_groups + 2, 0);
    for (size_t p = n_spheres; p < n; ++p) g_start[group_of[p] + 1]++;
    for (uint32_t g = 0; g <= n_groups; ++g) g_start[g + 1] += g_start[g];           // group g owns members[g_start[g] .. g_start[g+1])
    std::vector<uint32_t> members(n - n_spheres), fill(g_start.begin(), g_start.end() - 1);
    for (size_t p = n_spheres; p < n; ++p) members[fill[group_of[p]]++] = (uint32_t)p;
    const size_t n_loose_tri = g_start[1] - g_start[0];
    // ---- the groups' trees, side by side
    std::vector<BvhResult> sub(n_groups + 1);
    std::vector<PrimBox> gbox(n_groups + 1);
    {
        Team crew(build_threads());
        crew.run(n_groups, [&](size_t k) {
            const uint32_t g = (uint32_t)k + 1;
            const size_t m = g_start[g + 1] - g_start[g];
            std::vector<PrimBox> local(m);
            PrimBox b;
            for (int c = 0; c < 3; ++c) { b.lo[c] = INF; b.hi[c] = -INF; }
            for (size_t i = 0; i < m; ++i) {
                local[i] = boxes[members[g_start[g] + i]];
                for (int c = 0; c < 3; ++c) { b.lo[c] = std::min(b.lo[c], local[i].lo[c]); b.hi[c] = std::max(b.hi[c], local[i].hi[c]); }
            }
            gbox[g] = b;
            build_bvh_solo(local, 0, sub[g], SIZE_MAX, true);
            for (uint32_t& t : sub[g].tri_order) t = members[g_start[g] + t] - (uint32_t)n_spheres;      // back to the caller's triangle index
        });
    }
    // ---- the top tree: spheres, loose triangles, then one solo item per group
    std::vector<PrimBox> items(n_spheres + n_loose_tri + n_groups);
    for (size_t i = 0; i < n_spheres; ++i) items[i] = boxes[i];
    for (size_t i = 0; i < n_loose_tri; ++i) items[n_spheres + i] = boxes[members[g_start[0] + i]];
    for (uint32_t g = 1; g <= n_groups; ++g) items[n_spheres + n_loose_tri + g - 1] = gbox[g];
    BvhResult top;
    build_bvh_solo(items, n_spheres, top, n_spheres + n_loose_tri, false);
    // ---- stitch.  Final triangle order: the top tree's loose triangles in its leaf order, then group after group.
    out = BvhResult();
    out.sph_order = top.sph_order;
    std::vector<uint32_t> new_pos(top.tri_order.size(), 0xFFFFFFFFu);
    out.tri_order.reserve(n - n_spheres);
    for (size_t k = 0; k < top.tri_order.size(); ++k) {
        if (top.tri_order[k] < n_loose_tri) {
            new_pos[k] = (uint32_t)out.tri_order.size();
            out.tri_order.push_back(members[g_start[0] + top.tri_order[k]] - (uint32_t)n_spheres);
        }
    }
    std::vector<uint32_t> node_base(n_groups + 2, 0), tri_base(n_groups + 2, 0);
    node_base[1] = (uint32_t)top.nodes.size();
    tri_base[1] = (uint32_t)out.tri_order.size();
    for (uint32_t g = 1; g <= n_groups; ++g) {
        node_base[g + 1] = node_base[g] + (uint32_t)sub[g].nodes.size();
        tri_base[g + 1] = tri_base[g] + (uint32_t)sub[g].tri_order.size();
    }
    out.nodes.resize(node_base[n_groups + 1]);
    out.tri_order.resize(tri_base[n_groups + 1]);
    out.max_leaf = top.max_leaf;
    // the top tree's references: loose leaves move to their new positions, a group's solo leaf becomes its tree's root
    std::vector<uint32_t> group_depth(n_groups + 1, 0);
    {
        struct It { uint32_t node, depth; };
        std::vector<It> st;
        st.push_back({0u, 0u});
        auto fix = [&](uint32_t ref, uint32_t depth) -> uint32_t {
            if (ref == REF_EMPTY || !(ref & REF_LEAF)) { if (ref != REF_EMPTY) st.push_back({ref, depth}); return ref; }
            if (ref & REF_SPHERE) return ref;
            const uint32_t first = ref & REF_FIRST_MASK;
            const uint32_t item = top.tri_order[first];
            if (item >= n_loose_tri) {                                   // (alone in its leaf by construction)
                const uint32_t g = item - (uint32_t)n_loose_tri + 1;
                group_depth[g] = depth;
                return node_base[g];
            }
            return (ref & ~REF_FIRST_MASK) | new_pos[first];
        };
        while (!st.empty()) {
            const It it = st.back();
            st.pop_back();
            Node64 nd = top.nodes[it.node];
            nd.left = fix(nd.left, it.depth + 1);
            nd.right = fix(nd.right, it.depth + 1);
            out.nodes[it.node] = nd;
        }
    }
    out.depth = top.depth;
    {
        Team crew(build_threads());
        crew.run(n_groups, [&](size_t k) {
            const uint32_t g = (uint32_t)k + 1;
            const uint32_t nb = node_base[g], tb = tri_base[g];
            for (size_t i = 0; i < sub[g].nodes.size(); ++i) {
                Node64 nd = sub[g].nodes[i];
                auto fix = [&](uint32_t ref) -> uint32_t {
                    if (ref == REF_EMPTY) return ref;
                    return (ref & REF_LEAF) ? ref + tb : ref + nb;
                };
                nd.left = fix(nd.left);
                nd.right = fix(nd.right);
                out.nodes[nb + i] = nd;
            }
            std::copy(sub[g].tri_order.begin(), sub[g].tri_order.end(), out.tri_order.begin() + tb);
        });
    }
    for (uint32_t g = 1; g <= n_groups; ++g) {
        // a group's root node sits where the solo leaf's reference sat (depth d): its own references are d deeper
        out.depth = std::max(out.depth, group_depth[g] + sub[g].depth);
        out.max_leaf = std::max(out.max_leaf, sub[g].max_leaf);
    }
}

// node arrays are converted in pieces of this many nodes, by the build threads when the array is large
constexpr size_t NODE_PIECE = 32768;
static unsigned node_threads(size_t n_nodes) { return n_nodes >= 4 * NODE_PIECE ? build_threads() : 1u; }

void narrow_nodes(const std::vector<Node64>& in, std::vector<Node32>& out) {
    out.resize(in.size());
    Team crew(node_threads(in.size()));
    crew.run(pieces_of(in.size(), NODE_PIECE), [&](size_t piece) {
        for (size_t i = piece * NODE_PIECE; i < std::min(in.size(), (piece + 1) * NODE_PIECE); ++i) {
            const Node64& a = in[i];
            Node32& b = out[i];
            for (int k = 0; k < 3; ++k) {
                b.lo_l[k] = round_down(a.lo_l[k]); b.hi_l[k] = round_up(a.hi_l[k]);
                b.lo_r[k] = round_down(a.lo_r[k]); b.hi_r[k] = round_up(a.hi_r[k]);
            }
            b.left = a.left;
            b.right = a.right;
            b._pad[0] = b._pad[1] = 0;
        }
    });
}

bool quantize_nodes(const std::vector<Node32>& in, std::vector<Node16>& out, QuantGrid& g, double max_inflation) {
    if (in.empty()) return false;
    float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
    const size_t n_pieces = pieces_of(in.size(), NODE_PIECE);
    Team crew(node_threads(in.size()));
    {
        struct Ext { float lo[3], hi[3]; };
        std::vector<Ext> part(n_pieces);
        crew.run(n_pieces, [&](size_t piece) {
            Ext e = {{INFINITY, INFINITY, INFINITY}, {-INFINITY, -INFINITY, -INFINITY}};
            for (size_t i = piece * NODE_PIECE; i < std::min(in.size(), (piece + 1) * NODE_PIECE); ++i) {
                const Node32& n = in[i];
                for (int k = 0; k < 3; ++k) {
                    e.lo[k] = std::min(e.lo[k], n.lo_l[k]); e.hi[k] = std::max(e.hi[k], n.hi_l[k]);
                    if (n.right != REF_EMPTY) { e.lo[k] = std::min(e.lo[k], n.lo_r[k]); e.hi[k] = std::max(e.hi[k], n.hi_r[k]); }
                }
            }
            part[piece] = e;
        });
        for (const Ext& e : part)
            for (int k = 0; k < 3; ++k) { lo[k] = std::min(lo[k], e.lo[k]); hi[k] = std::max(hi[k], e.hi[k]); }
    }
    for (int k = 0; k < 3; ++k) {
        if (!std::isfinite(lo[k]) || !std::isfinite(hi[k])) return false;
        g.origin[k] = lo[k];
        double cell = ((double)hi[k] - (double)lo[k]) / 65535.0 * 1.0001;
        g.cell[k] = round_up(cell > 0.0 ? cell : 1e-30);
    }
    // the plane a kernel tests against is the REAL number origin + q*cell (rays are moved into cell units,
    // q itself is the coordinate): 16 x 24-bit product plus a 24-bit term, exact in 64-bit-mantissa arithmetic
    auto dec = [&](uint32_t q, int k) { return (long double)g.origin[k] + (long double)q * (long double)g.cell[k]; };
    auto qlo = [&](float v, int k, uint16_t& o) {
        double q = std::floor(((double)v - (double)g.origin[k]) / (double)g.cell[k]);
        long qi = (long)std::min(std::max(q, 0.0), 65535.0);
        while (qi > 0 && dec((uint32_t)qi, k) > v) --qi;
        if (dec((uint32_t)qi, k) > v) return false;
        o = (uint16_t)qi;
        return true;
    };
    auto qhi = [&](float v, int k, uint16_t& o) {
        double q = std::ceil(((double)v - (double)g.origin[k]) / (double)g.cell[k]);
        long qi = (long)std::min(std::max(q, 0.0), 65535.0);
        while (qi < 65535 && dec((uint32_t)qi, k) < v) ++qi;
        if (dec((uint32_t)qi, k) < v) return false;
        o = (uint16_t)qi;
        return true;
    };
    std::vector<Node16> res(in.size());
    auto area = [](const float* l, const float* h, const double pad[3]) {
        double e[3];
        for (int k = 0; k < 3; ++k) e[k] = (double)h[k] - (double)l[k] + 2.0 * pad[k];
        return e[0] * e[1] + e[1] * e[2] + e[2] * e[0];
    };
    const double pad0[3] = {0, 0, 0}, pad1[3] = {g.cell[0], g.cell[1], g.cell[2]};
    // per piece: the inflation sum of its leaf boxes, their number, and whether every plane could be placed; the sums
    // are added up in piece order, so the decision below does not depend on the thread count
    struct Part { double infl; size_t n_leaf; bool ok; };
    std::vector<Part> part(n_pieces);
    crew.run(n_pieces, [&](size_t piece) {
        Part pt = {0.0, 0, true};
        for (size_t i = piece * NODE_PIECE; i < std::min(in.size(), (piece + 1) * NODE_PIECE) && pt.ok; ++i) {
            const Node32& a = in[i];
            Node16& b = res[i];
            const bool has_r = a.right != REF_EMPTY;
            for (int k = 0; k < 3 && pt.ok; ++k) {
                if (!qlo(a.lo_l[k], k, b.lo_l[k]) || !qhi(a.hi_l[k], k, b.hi_l[k])) pt.ok = false;
                if (has_r) {
                    if (!qlo(a.lo_r[k], k, b.lo_r[k]) || !qhi(a.hi_r[k], k, b.hi_r[k])) pt.ok = false;
                } else {
                    b.lo_r[k] = 65535; b.hi_r[k] = 0;
                }
            }
            b.left = a.left;
            b.right = a.right;
            if (a.left & REF_LEAF) { double s0 = area(a.lo_l, a.hi_l, pad0); if (s0 > 0) { pt.infl += area(a.lo_l, a.hi_l, pad1) / s0; ++pt.n_leaf; } }
            if (has_r && (a.right & REF_LEAF)) { double s0 = area(a.lo_r, a.hi_r, pad0); if (s0 > 0) { pt.infl += area(a.lo_r, a.hi_r, pad1) / s0; ++pt.n_leaf; } }
        }
        part[piece] = pt;
    });
    double infl = 0.0;
    size_t n_leaf = 0;
    for (const Part& pt : part) {
        if (!pt.ok) return false;
        infl += pt.infl;
        n_leaf += pt.n_leaf;
    }
    if (n_leaf && infl / (double)n_leaf > max_inflation) return false;
    out.swap(res);
    return true;
}

namespace {
template <class NodeT>
struct Wide {
    const std::vector<Node64>& in;
    const QuantGrid& g;
    std::vector<NodeT>& out;
    bool ok = true;
    struct Ent { uint32_t ref; double lo[3], hi[3]; };
    long double dec(uint32_t q, int k) const { return (long double)g.origin[k] + (long double)q * (long double)g.cell[k]; }   // exact plane
    uint16_t qlo(double x, int k) {
        const float v = round_down(x);
        double q = std::floor(((double)v - (double)g.origin[k]) / (double)g.cell[k]);
        long qi = (long)std::min(std::max(q, 0.0), 65535.0);
        while (qi > 0 && dec((uint32_t)qi, k) > v) --qi;
        if (dec((uint32_t)qi, k) > v) ok = false;
        return (uint16_t)qi;
    }
    uint16_t qhi(double x, int k) {
        const float v = round_up(x);
        double q = std::ceil(((double)v - (double)g.origin[k]) / (double)g.cell[k]);
        long qi = (long)std::min(std::max(q, 0.0), 65535.0);
        while (qi < 65535 && dec((uint32_t)qi, k) < v) ++qi;
        if (dec((uint32_t)qi, k) < v) ok = false;
        return (uint16_t)qi;
    }
    static double area(const Ent& e) {
        const double x = e.hi[0] - e.lo[0], y = e.hi[1] - e.lo[1], z = e.hi[2] - e.lo[2];
        return x * y + y * z + z * x;
    }
    static void children(const Node64& n, Ent* e, int& c) {
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
    // ---- which BVH2 nodes become BVH4 nodes: the optimal choice for the surface-area cost (sum of the BVH4 nodes' areas:
    // the leaves are the BVH2's own and cost the same whatever is chosen), by the dynamic programme of Ylitie, Karras and
    // Laine ("Efficient incoherent ray traversal on GPUs through compressed wide BVHs", 2017, section 3.1) for width 4.
    // plan[3 m + (k - 2)], k = 2, 3, 4: how many of the k roots that may stand for BVH2 node m's subtree go to its LEFT
    // child; 0 (k = 2, 3 only) = m stays one root.
    const uint8_t* plan = nullptr;
    static void make_plan(const std::vector<Node64>& in, std::vector<uint8_t>& plan) {
        const size_t n = in.size();
        plan.assign(3 * n, 0);
        std::vector<double> cost(3 * n, 0.0);                    // cost[3 m + (k - 1)]: m's subtree as at most k roots, k = 1..3
        auto of = [&](uint32_t ref, int k) -> double {          // (a leaf costs the same in every plan: 0)
            return (ref & REF_LEAF) ? 0.0 : cost[3 * (size_t)ref + (size_t)(std::min(k, 3) - 1)];
        };
        for (size_t m = n; m-- > 0;) {                           // children sit behind their parents
            const Node64& nd = in[m];
            double lo[3], hi[3];
            for (int k = 0; k < 3; ++k) { lo[k] = nd.lo_l[k]; hi[k] = nd.hi_l[k]; }
            if (nd.right != REF_EMPTY)
                for (int k = 0; k < 3; ++k) { lo[k] = std::min(lo[k], nd.lo_r[k]); hi[k] = std::max(hi[k], nd.hi_r[k]); }
            const double x = hi[0] - lo[0], y = hi[1] - lo[1], z = hi[2] - lo[2];
            const double area_m = x * y + y * z + z * x;
            double dist[5] = {0, 0, 0, 0, 0};
            uint8_t arg[5] = {0, 0, 0, 0, 0};
            for (int k = 2; k <= 4; ++k) {
                if (nd.right == REF_EMPTY) { dist[k] = of(nd.left, k); arg[k] = (uint8_t)std::min(k, 3); continue; }
                double best = INFINITY;
                for (int i = std::max(1, k - 3); i <= std::min(3, k - 1); ++i) {
                    const double c = of(nd.left, i) + of(nd.right, k - i);
                    if (c < best) { best = c; arg[k] = (uint8_t)i; }
                }
                dist[k] = best;
            }
            const double c1 = area_m + dist[4];
            const double c2 = std::min(c1, dist[2]);
            const double c3 = std::min(c2, dist[3]);
            cost[3 * m] = c1; cost[3 * m + 1] = c2; cost[3 * m + 2] = c3;
            plan[3 * m] = dist[2] < c1 ? arg[2] : 0;
            plan[3 * m + 1] = dist[3] < c2 ? arg[3] : 0;
            plan[3 * m + 2] = arg[4];
        }
    }
    // the roots (at most k) that stand for the child `side` of `parent` under the plan
    void expand(const Node64& parent, bool right_side, int k, Ent* e, int& c) const {
        Ent x;
        x.ref = right_side ? parent.right : parent.left;
        for (int j = 0; j < 3; ++j) { x.lo[j] = right_side ? parent.lo_r[j] : parent.lo_l[j]; x.hi[j] = right_side ? parent.hi_r[j] : parent.hi_l[j]; }
        while (!(x.ref & REF_LEAF) && k > 1) {
            const Node64& m = in[x.ref];
            const int i = plan[3 * (size_t)x.ref + (size_t)(k - 2)];
            if (i == 0) { --k; continue; }                      // the plan for one root fewer
            if (m.right == REF_EMPTY) { expand(m, false, i, e, c); return; }
            expand(m, false, i, e, c);
            expand(m, true, k - i, e, c);
            return;
        }
        e[c++] = x;
    }
    // the children of the BVH4 node that replaces BVH2 node n2
    int collapse(uint32_t n2, Ent* e) const {
        int c = 0;
#ifndef RT_WIDEN_GREEDY
        if (plan) {
            const Node64& nd = in[n2];
            const int i = plan[3 * (size_t)n2 + 2];
            expand(nd, false, i, e, c);
            if (nd.right != REF_EMPTY) expand(nd, true, 4 - i, e, c);
            return c;
        }
#endif
        // (without a plan: its own two, then the grandchildren with the largest boxes pulled up until there are four)
        children(in[n2], e, c);
        while (c < 4) {
            int best = -1;
            double best_area = -1.0;
            for (int i = 0; i < c; ++i)
                if (!(e[i].ref & REF_LEAF) && area(e[i]) > best_area) { best = i; best_area = area(e[i]); }
            if (best < 0) break;
            Ent sub[2];
            int cs = 0;
            children(in[e[best].ref], sub, cs);
            e[best] = sub[0];
            if (cs == 2) e[c++] = sub[1];
        }
        return c;
    }
    // number of BVH4 nodes the subtree of n2 becomes
    size_t count(uint32_t n2) const {
        Ent e[5];
        const int c = collapse(n2, e);
        size_t k = 1;
        for (int i = 0; i < c; ++i)
            if (!(e[i].ref & REF_LEAF)) k += count(e[i].ref);
        return k;
    }
    // Emits the BVH4 subtree that replaces BVH2 node n2 into out[cursor...] in pre-order (a node, then the subtrees of
    // its children one after the other); out is sized beforehand.
    void emit(uint32_t n2, uint32_t& cursor) {
        Ent e[5];
        const int c = collapse(n2, e);
        const uint32_t index = cursor++;
        uint32_t refs[4] = {REF_EMPTY, REF_EMPTY, REF_EMPTY, REF_EMPTY};
        for (int i = 0; i < c; ++i) {
            if (e[i].ref & REF_LEAF) { refs[i] = e[i].ref; continue; }
            refs[i] = cursor;
            emit(e[i].ref, cursor);
        }
        store(out[index], e, c, refs);
    }
    // The subtrees that hang `levels` BVH4 levels below n2, in the order emit reaches them.
    void frontier(uint32_t n2, int levels, std::vector<uint32_t>& roots) const {
        if (levels == 0) { roots.push_back(n2); return; }
        Ent e[5];
        const int c = collapse(n2, e);
        for (int i = 0; i < c; ++i)
            if (!(e[i].ref & REF_LEAF)) frontier(e[i].ref, levels - 1, roots);
    }
    // emit for the nodes above the frontier: same order, the frontier subtrees only get their place (base[k], sizes
    // size[k], consumed in order through `next`)
    void emit_top(uint32_t n2, int levels, uint32_t& cursor, const std::vector<size_t>& size, std::vector<uint32_t>& base, size_t& next) {
        if (levels == 0) {
            base[next] = cursor;
            cursor += (uint32_t)size[next];
            ++next;
            return;
        }
        Ent e[5];
        const int c = collapse(n2, e);
        const uint32_t index = cursor++;
        uint32_t refs[4] = {REF_EMPTY, REF_EMPTY, REF_EMPTY, REF_EMPTY};
        for (int i = 0; i < c; ++i) {
            if (e[i].ref & REF_LEAF) { refs[i] = e[i].ref; continue; }
            refs[i] = cursor;
            emit_top(e[i].ref, levels - 1, cursor, size, base, next);
        }
        store(out[index], e, c, refs);
    }
    void store(Node4& nd, const Ent* e, int c, const uint32_t* refs) {
        for (int i = 0; i < 4; ++i) {
            nd.child[i] = refs[i];
            for (int k = 0; k < 3; ++k) {
                nd.box[i][k] = i < c ? qlo(e[i].lo[k], k) : (uint16_t)65535;
                nd.box[i][3 + k] = i < c ? qhi(e[i].hi[k], k) : (uint16_t)0;
            }
        }
    }
    void store(Node4f& nd, const Ent* e, int c, const uint32_t* refs) {
        for (int i = 0; i < 4; ++i) {
            nd.child[i] = refs[i];
            nd._pad[i] = 0;
            for (int k = 0; k < 3; ++k) {
                nd.box[i][k] = i < c ? round_down(e[i].lo[k]) : INFINITY;
                nd.box[i][3 + k] = i < c ? round_up(e[i].hi[k]) : -INFINITY;
            }
        }
    }
};
}  // namespace

namespace {
// BVH2 -> BVH4 for either node format.  The subtrees a few levels below the root are counted, then emitted, by the
// build threads side by side; where each lands follows from the counts, so the result is the pre-order numbering a
// single depth-first emit produces, whatever the thread count.
template <class NodeT>
bool widen_any(const std::vector<Node64>& in, const QuantGrid& grid, std::vector<NodeT>& out, uint32_t& stack_need) {
    if (in.empty()) return false;
    std::vector<NodeT> res;
    Wide<NodeT> top{in, grid, res};
    std::vector<uint8_t> plan;
    Wide<NodeT>::make_plan(in, plan);
    top.plan = plan.data();
    const unsigned T = build_threads();
    const int levels = (T > 1 && in.size() >= 100000) ? 4 : 0;
    std::vector<uint32_t> roots;
    top.frontier(0, levels, roots);
    std::vector<size_t> size(roots.size());
    std::vector<uint32_t> base(roots.size());
    Team crew(levels ? T : 1u);
    crew.run(roots.size(), [&](size_t k) { size[k] = top.count(roots[k]); });
    size_t total = 0;
    for (size_t k = 0; k < roots.size(); ++k) total += size[k];
    total += 1024;                                  // the nodes above the frontier: at most 1 + 4 + 16 + 64
    if (total >= (size_t)REF_FIRST_MASK) return false;
    res.resize(total);
    uint32_t cursor = 0;
    size_t next = 0;
    top.emit_top(0, levels, cursor, size, base, next);
    res.resize(cursor);
    std::vector<char> bad(roots.size(), 0);
    crew.run(roots.size(), [&](size_t k) {
        Wide<NodeT> w{in, grid, res};
        w.plan = plan.data();
        uint32_t c = base[k];
        w.emit(roots[k], c);
        bad[k] = (!w.ok || c != base[k] + size[k]) ? 1 : 0;
    });
    if (!top.ok) return false;
    for (char f : bad)
        if (f) return false;
    // stack need of a near-first traversal that pushes every other hit child: children sit behind their parent, so one
    // backward sweep has every child's need before its parent's
    std::vector<uint8_t> need(res.size(), 0);
    for (size_t i = res.size(); i-- > 0;) {
        uint32_t c = 0, deepest = 0;
        for (int k = 0; k < 4; ++k) {
            const uint32_t ref = res[i].child[k];
            if (ref == REF_EMPTY) continue;
            ++c;
            if (!(ref & REF_LEAF)) deepest = std::max<uint32_t>(deepest, need[ref]);
        }
        need[i] = (uint8_t)std::min<uint32_t>(255u, (c ? c - 1 : 0) + deepest);
    }
    stack_need = need[0];
    out.swap(res);
    return true;
}
}  // namespace

bool widen_nodes(const std::vector<Node64>& in, const QuantGrid& grid, std::vector<Node4>& out, uint32_t& stack_need) {
    return widen_any(in, grid, out, stack_need);
}

bool widen_nodes_f32(const std::vector<Node64>& in, std::vector<Node4f>& out, uint32_t& stack_need) {
    const QuantGrid none{};
    return widen_any(in, none, out, stack_need);
}

}  // namespace rt
