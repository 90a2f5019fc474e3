// rt_kernels.hip — wavefront path-tracing kernels for gfx950 (wave64).
//
//   K0 k_raygen        Camera::get_ray                     src/camera.rs:231-254, :367-371, :140-165
//   K1 k_walk          World::hit over the BVH             src/hittable.rs:135-149, :318-365, :408-495
//      k_trace_packet  the same for coherent camera rays (+ K0 and depth-0 shading fused in)
//   K2-4 k_shade       sky on miss, Material::scatter,     src/hittable.rs:84-93, src/material.rs:94-179,
//                      russian_roulette, compaction        src/camera.rs:280-313
//      k_walk_shade    K1 + K2-4 in one persistent kernel: a lane traces, shades and goes on with the scattered
//                      ray until its path ends, then takes the next path from the queue (the raycast recursion of
//                      src/camera.rs:296-313 unrolled per lane)
//   K5 k_accumulate / k_resolve   render_pixel's mean      src/camera.rs:315-325
//
// One lane = one path.  Between bounces the survivors are compacted into the next queue with a
// wave64 ballot + one atomic per wave; every consumer reads the queue's entry counts from device
// memory, so the host queues the kernels of a whole pass without a round trip.  Radiance is written
// once per path into its own slot and summed per pixel in sample order, so a render is bit-reproducible.
#include "rt_kernels.h"

#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <type_traits>

#include "rt_device.h"

namespace rt {

// ---------------------------------------------------------------------------------------------
// K0 — primary rays
// ---------------------------------------------------------------------------------------------
// Camera::get_ray for the path with index `ix` of the pass (src/camera.rs:231-254)
__device__ __forceinline__ void camera_ray(const PassParams& pp, const PathIndex& ix, D3& origin, D3& dir) {
    const uint32_t pixel = ix.pixel, x = ix.x, y = ix.y;
    const uint32_t sample = pp.sample_offset + (pp.pass_base + ix.local) * pp.sample_stride;
    // rng_map[i] (src/camera.rs:239): from the pass's table when the host provided one (exact rationals either way)
    double ox = 0.0, oy = 0.0;
    if (pp.jitter) {
        if (pp.jitter_table) { ox = pp.jitter_table[2 * ix.local]; oy = pp.jitter_table[2 * ix.local + 1]; }
        else { ox = halton2(sample); oy = halton3(sample); }
    }
    D3 p00 = ld3(pp.cam.pixel00), du = ld3(pp.cam.pixel_du), dv = ld3(pp.cam.pixel_dv), center = ld3(pp.cam.center);
    D3 ps = add(add(p00, scale(du, (double)x + ox)), scale(dv, (double)y + oy));
    origin = center;
    if (pp.jitter && !(pp.cam.defocus_angle <= 0.0)) {
        // Vec3::random_in_unit_disc (src/vec3.rs:124-135); rejection k uses Philox block k of the
        // primary-ray key (depth tag 0xFFFFFFFF); capped at 32 rejections like the oracle
        RngKey key = {pp.seed, pixel, sample, DEPTH_PRIMARY};
        D3 v = mk(1.0, 1.0, 1.0);
        for (uint32_t it = 0; it < 32 && norm2(v) > 1.0; ++it) {
            U4 r = rng_block(key, it);
            v = mk(2.0 * u64_to_unit(r.x, r.y) - 1.0, 2.0 * u64_to_unit(r.z, r.w) - 1.0, 0.0);
        }
        origin = add(add(center, scale(ld3(pp.cam.defocus_u), v.x)), scale(ld3(pp.cam.defocus_v), v.y));
    }
    dir = normalize(sub(ps, origin));                    // Ray::new normalises
}

// rng_map of one pass (src/camera.rs:208-210): the Halton (2, 3) pair of every local sample, computed once instead of once
// per camera ray (the base-3 radical inverse is a loop of divisions).  out[2 * local] = H2, out[2 * local + 1] = H3.
__global__ __launch_bounds__(BLOCK) void k_jitter_table(uint32_t sample_offset, uint32_t sample_stride, uint32_t pass_base, uint32_t n_local,
                                                        double* out) {
    const uint32_t local = blockIdx.x * BLOCK + threadIdx.x;
    if (local >= n_local) return;
    const uint32_t sample = sample_offset + (pass_base + local) * sample_stride;
    out[2 * local] = halton2(sample);
    out[2 * local + 1] = halton3(sample);
}

__global__ __launch_bounds__(BLOCK) void k_raygen(PassParams pp, QueueGeom qg, RayQueue q) {
    const uint32_t pid = blockIdx.x * BLOCK + threadIdx.x;       // logical entry = path id
    uint32_t slot = 0;
    if (!direct_slot(qg, pid, slot)) return;
    D3 origin, dir;
    camera_ray(pp, decode_path(pp, pid), origin, dir);
    q.o[0][slot] = origin.x; q.o[1][slot] = origin.y; q.o[2][slot] = origin.z;
    q.d[0][slot] = dir.x; q.d[1][slot] = dir.y; q.d[2][slot] = dir.z;
    q.pid[slot] = pid;                                   // (throughput starts at 1: k_shade does not read it at depth 0)
}

__global__ __launch_bounds__(BLOCK) void k_prepare_rays(const rt_ray* rays, QueueGeom qg, RayQueue q) {
    const uint32_t p = blockIdx.x * BLOCK + threadIdx.x;
    uint32_t i = 0;
    if (!direct_slot(qg, p, i)) return;
    D3 o = ld3(rays[p].o), d = normalize(ld3(rays[p].d));
    q.o[0][i] = o.x; q.o[1][i] = o.y; q.o[2][i] = o.z;
    q.d[0][i] = d.x; q.d[1][i] = d.y; q.d[2][i] = d.z;
    q.pid[i] = p;
}

// ---------------------------------------------------------------------------------------------
// K1 — closest hit.  Per-lane traversal stack in LDS ([entry][lane]: conflict-free).
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ double mkd(uint32_t lo, uint32_t hi) { return __hiloint2double((int)hi, (int)lo); }

// Ray-side state of the slab test, one policy per node format.
//
// Box64: f64 boxes.  The interval is widened by a few ulps so that a primitive touching its own
// (exact) box is never culled by rounding; NaNs from 0*inf drop out of fmin/fmax.
struct Box64 {
    using Node = Node64;
    D3 o, inv;
    double lo_t, hi_t;     // [t_min, current nearest t]
    __device__ __forceinline__ void init(const DScene&, D3 o_, D3 d, double t_min, double t_max) {
        o = o_; inv = mk(1.0 / d.x, 1.0 / d.y, 1.0 / d.z); lo_t = t_min; hi_t = t_max;
    }
    __device__ __forceinline__ void nearest(double best) { hi_t = best; }
    using Scalar = double;
    __device__ __forceinline__ void fetch(const DScene& S, uint32_t cur, double* lo_l, double* hi_l, double* lo_r, double* hi_r,
                                          uint32_t& cl, uint32_t& cr) const {
        const Node64* nd = &S.nodes64[cur];
        for (int k = 0; k < 3; ++k) { lo_l[k] = nd->lo_l[k]; hi_l[k] = nd->hi_l[k]; lo_r[k] = nd->lo_r[k]; hi_r[k] = nd->hi_r[k]; }
        cl = nd->left; cr = nd->right;
    }
    __device__ __forceinline__ bool test(const double* lo, const double* hi, double& t_near) const {
        double ax = (lo[0] - o.x) * inv.x, bx = (hi[0] - o.x) * inv.x;
        double ay = (lo[1] - o.y) * inv.y, by = (hi[1] - o.y) * inv.y;
        double az = (lo[2] - o.z) * inv.z, bz = (hi[2] - o.z) * inv.z;
        double tn = fmax(fmax(fmin(ax, bx), fmin(ay, by)), fmin(az, bz));
        double tf = fmin(fmin(fmax(ax, bx), fmax(ay, by)), fmax(az, bz));
        tn *= tn > 0.0 ? 1.0 - 1e-15 : 1.0 + 1e-15;      // (multiplying keeps +-inf; subtracting would make NaN)
        tf *= tf > 0.0 ? 1.0 + 1e-15 : 1.0 - 1e-15;
        t_near = tn;
        return !(tn > tf) && !(tf < lo_t) && !(tn > hi_t);
    }
};

// Box32: conservative f32 test against boxes rounded OUTWARD to f32 (bvh_build.cpp: narrow_nodes).
// It only culls; every accepted primitive is still intersected in f64, so (prim, t) stay bit-exact.
// The kernels are bound by VALU issue in this test, so it is written as one fma per slab plane:
//     Y = fma(b, idf, KN)            idf = fl32(1/d),  K = fl32(o * idf),  c = |K| 2^-21,  KN = fl32(-K - c)
// (KF = fl32(-K + c) for the far planes).  Error budget, u = 2^-24, against the exact t = (b - o)/d:
//     b*idf - K = (1 + e2)[t - A e3],   A = o/d,  |e2|, |e3| <= u          (rounding of idf and of K)
//     Y = (b*idf - K - c + r)(1 + e4),  |r| <= u(|K| + c), |e4| <= u       (rounding of KN and of the fma)
//     =>  t - Y >= -2.1u |Y| + (c - 3.1u |K|)  >=  -2.1u |Y|               (and symmetrically for KF)
// so  Y - |Y| 2^-22  <= t for near planes and  Y + |Y| 2^-22 >= t  for far planes; y -> y -/+ |y| 2^-22 is
// monotone and the same for the three axes, so it is applied once after max3 / min3.  0*inf and inf-inf
// NaNs (rays parallel to a slab) drop out of fminf/fmaxf (IEEE minNum/maxNum), which only ever widens
// the interval; the final comparisons are written negated so that a NaN accepts.
struct Box32 {
    using Node = Node32;
    float ix, iy, iz, nx, ny, nz, fx, fy, fz;      // idf, KN, KF per axis
    float lo_t, hi_t;      // f32 bracket of [t_min, current nearest t]: lo_t <= t_min, hi_t >= nearest
    static __device__ __forceinline__ float f32_below(double x) { float f = (float)x; return f - fabsf(f) * 0x1.0p-22f; }
    static __device__ __forceinline__ float f32_above(double x) { float f = (float)x; return f + fabsf(f) * 0x1.0p-22f; }
    __device__ __forceinline__ void nearest(double best) { hi_t = f32_above(best); }
    // o, inv in the space the boxes are given in (world for Box32, grid cells for Box16)
    __device__ __forceinline__ void set_axis(double o, double inv, float& idf, float& kn, float& kf) {
        idf = (float)inv;
        const float K = (float)(o * (double)idf);
        const float c = fmaf(fabsf(K), 0x1.0p-21f, 1e-30f);
        kn = -K - c;
        kf = -K + c;
    }
    __device__ __forceinline__ void init(const DScene&, D3 o, D3 d, double t_min, double t_max) {
        lo_t = f32_below(t_min); hi_t = f32_above(t_max);
        set_axis(o.x, 1.0 / d.x, ix, nx, fx);
        set_axis(o.y, 1.0 / d.y, iy, ny, fy);
        set_axis(o.z, 1.0 / d.z, iz, nz, fz);
    }
    using Scalar = float;
    __device__ __forceinline__ void fetch(const DScene& S, uint32_t cur, float* lo_l, float* hi_l, float* lo_r, float* hi_r,
                                          uint32_t& cl, uint32_t& cr) const {
        const Node32* nd = &S.nodes32[cur];
        for (int k = 0; k < 3; ++k) { lo_l[k] = nd->lo_l[k]; hi_l[k] = nd->hi_l[k]; lo_r[k] = nd->lo_r[k]; hi_r[k] = nd->hi_r[k]; }
        cl = nd->left; cr = nd->right;
    }
    // the widened parametric interval [tn, tf] of the ray inside a box; accept() is the test on it
    __device__ __forceinline__ void interval(const float* lo, const float* hi, float& tn, float& tf) const {
        const float r = 0x1.0p-22f;
        const float ax = fminf(fmaf(lo[0], ix, nx), fmaf(hi[0], ix, nx)), bx = fmaxf(fmaf(lo[0], ix, fx), fmaf(hi[0], ix, fx));
        const float ay = fminf(fmaf(lo[1], iy, ny), fmaf(hi[1], iy, ny)), by = fmaxf(fmaf(lo[1], iy, fy), fmaf(hi[1], iy, fy));
        const float az = fminf(fmaf(lo[2], iz, nz), fmaf(hi[2], iz, nz)), bz = fmaxf(fmaf(lo[2], iz, fz), fmaf(hi[2], iz, fz));
        tn = fmaxf(fmaxf(ax, ay), az);
        tf = fminf(fminf(bx, by), bz);
        tn = fmaf(-fabsf(tn), r, tn);
        tf = fmaf(fabsf(tf), r, tf);
    }
    __device__ __forceinline__ bool accept(float tn, float tf) const { return !(tn > tf) && !(tf < lo_t) && !(tn > hi_t); }
    // accept() for the whole wave as a lane mask, straight from its three compares (a NaN on either side accepts, as in
    // accept()); written with the compare-to-mask builtin because a ballot over a computed flag goes through a 0/1
    // register and a second compare
    __device__ __forceinline__ unsigned long long accept_mask(float tn, float tf) const {
        constexpr int UGE = 11;      // LLVM FCMP_UGE: unordered or >=
        return __builtin_amdgcn_fcmpf(tf, tn, UGE) & __builtin_amdgcn_fcmpf(tf, lo_t, UGE) & __builtin_amdgcn_fcmpf(hi_t, tn, UGE);
    }
    __device__ __forceinline__ bool test(const float* lo, const float* hi, float& t_near) const {
        float tn, tf;
        interval(lo, hi, tn, tf);
        t_near = tn;
        return accept(tn, tf);
    }
    // The same test with the near and far plane of each axis already picked by the sign of the direction (near = lo
    // when idf >= 0, hi otherwise): one fma per plane, no min/max per axis.  For finite planes and idf it returns what
    // test() returns (rounding is monotone); a ray parallel to a slab gives NaN for both planes of that axis in either
    // form, and the NaN drops out of max3 / min3 the same way.
    __device__ __forceinline__ void interval_picked(const float* near, const float* far, float& tn, float& tf) const {
        const float r = 0x1.0p-22f;
        tn = fmaxf(fmaxf(fmaf(near[0], ix, nx), fmaf(near[1], iy, ny)), fmaf(near[2], iz, nz));
        tf = fminf(fminf(fmaf(far[0], ix, fx), fmaf(far[1], iy, fy)), fmaf(far[2], iz, fz));
        tn = fmaf(-fabsf(tn), r, tn);
        tf = fmaf(fabsf(tf), r, tf);
    }
    __device__ __forceinline__ bool test_picked(const float* near, const float* far, float& t_near) const {
        float tn, tf;
        interval_picked(near, far, tn, tf);
        t_near = tn;
        return accept(tn, tf);
    }
};

// Box16: the same test in GRID space for nodes quantised to the global 16-bit grid (Node16, Node4): the
// ray is moved into cell units once (o' = (o - origin)/cell, 1/d' = cell/d, in f64 before the roundings the
// budget above accounts for), and a box plane is the integer q itself — exact in f32, no decode arithmetic
// per node.  The builder checks in exact arithmetic that origin + q*cell encloses every box.
struct Box16 : Box32 {
    using Node = Node16;
    // byte selectors for v_perm_b32: out of the three words of a Node4 child they build (near16 | far16 << 16) for
    // each axis, near = lo when the ray runs in +axis direction, hi otherwise
    uint32_t sx, sy, sz;
    __device__ __forceinline__ void init(const DScene& S, D3 o, D3 d, double t_min, double t_max) {
        lo_t = f32_below(t_min); hi_t = f32_above(t_max);
        // (o - origin) * (1/cell) instead of a division: a relative 2^-52 on o', against the 2^-21 |K| the planes are
        // widened by (budget above); three f64 divisions fewer per ray taken from the queue
        const double cx = (double)S.grid.cell[0], cy = (double)S.grid.cell[1], cz = (double)S.grid.cell[2];
        set_axis((o.x - (double)S.grid.origin[0]) * S.grid_inv_cell[0], cx / d.x, ix, nx, fx);
        set_axis((o.y - (double)S.grid.origin[1]) * S.grid_inv_cell[1], cy / d.y, iy, ny, fy);
        set_axis((o.z - (double)S.grid.origin[2]) * S.grid_inv_cell[2], cz / d.z, iz, nz, fz);
        sx = ix < 0.f ? 0x01000706u : 0x07060100u;      // lo.x = w0[15:0],  hi.x = w1[31:16]   (perm(w1, w0))
        sy = iy < 0.f ? 0x03020504u : 0x05040302u;      // lo.y = w0[31:16], hi.y = w2[15:0]    (perm(w2, w0))
        sz = iz < 0.f ? 0x01000706u : 0x07060100u;      // lo.z = w1[15:0],  hi.z = w2[31:16]   (perm(w2, w1))
    }
    // The slab test on one Node4 child (words w0, w1, w2) without min/max: the near and far plane of each axis are
    // picked by the sign of the direction (one v_perm_b32 per axis), then one fma per plane.  Same error budget as
    // Box32::test (the planes and the roundings are the same ones); a NaN (ray parallel to the slab) drops out of
    // fmaxf/fminf or reaches the negated comparisons, which accept.
    __device__ __forceinline__ bool test_child(uint32_t w0, uint32_t w1, uint32_t w2, float& t_near) const {
        const float r = 0x1.0p-22f;
        const uint32_t px = __builtin_amdgcn_perm(w1, w0, sx), py = __builtin_amdgcn_perm(w2, w0, sy), pz = __builtin_amdgcn_perm(w2, w1, sz);
        const float ax = fmaf((float)(px & 0xFFFFu), ix, nx), bx = fmaf((float)(px >> 16), ix, fx);
        const float ay = fmaf((float)(py & 0xFFFFu), iy, ny), by = fmaf((float)(py >> 16), iy, fy);
        const float az = fmaf((float)(pz & 0xFFFFu), iz, nz), bz = fmaf((float)(pz >> 16), iz, fz);
        float tn = fmaxf(fmaxf(ax, ay), az);
        float tf = fminf(fminf(bx, by), bz);
        tn = fmaf(-fabsf(tn), r, tn);
        tf = fmaf(fabsf(tf), r, tf);
        t_near = tn;
        return !(tn > tf) && !(tf < lo_t) && !(tn > hi_t);
    }
    __device__ __forceinline__ void fetch(const DScene& S, uint32_t cur, float* lo_l, float* hi_l, float* lo_r, float* hi_r,
                                          uint32_t& cl, uint32_t& cr) const {
        const uint4* p = reinterpret_cast<const uint4*>(&S.nodes16[cur]);
        const uint4 a = p[0], b = p[1];
        lo_l[0] = (float)(a.x & 0xFFFFu); lo_l[1] = (float)(a.x >> 16); lo_l[2] = (float)(a.y & 0xFFFFu);
        hi_l[0] = (float)(a.y >> 16);     hi_l[1] = (float)(a.z & 0xFFFFu); hi_l[2] = (float)(a.z >> 16);
        lo_r[0] = (float)(a.w & 0xFFFFu); lo_r[1] = (float)(a.w >> 16);     lo_r[2] = (float)(b.x & 0xFFFFu);
        hi_r[0] = (float)(b.x >> 16);     hi_r[1] = (float)(b.y & 0xFFFFu); hi_r[2] = (float)(b.y >> 16);
        cl = b.z; cr = b.w;
    }
};

// One wave per workgroup: a workgroup's LDS is held until its slowest wave retires, and ray
// lengths diverge a lot, so multi-wave groups strand LDS (and with it occupancy) behind stragglers.
constexpr int TRACE_BLOCK = 64;

#ifndef RT_TRACE_WAVES
#define RT_TRACE_WAVES 1
#endif

__device__ __forceinline__ bool shade_path(const DScene& S, const PassParams& pp, uint32_t depth, uint32_t pid, uint32_t pixel,
                                           uint32_t local, uint32_t i, const D3 o, const D3 d, const D3 thr, uint32_t prim, double t,
                                           const HitBuf& hb, double* path_radiance, D3& no, D3& nd, D3& nthr);
__device__ __forceinline__ void queue_append(bool alive, uint32_t wave, uint32_t lane, const RayQueue& qo, uint32_t shard_cap,
                                             uint32_t* out_counts, uint32_t pid, const D3 no, const D3 nd, const D3 nthr);

// K1 — per-lane traversal, the kernel of every incoherent ray.  WIDTH = 4 walks the BVH4 (Node4 on the 16-bit grid
// with BoxT = Box16; Node4f, 128-byte nodes with f32 boxes in world space, with BoxT = Box32, for scenes whose grid would
// be too coarse): a step fetches one node, tests its four child boxes, sorts the hit children by entry distance (5
// compare-exchanges), descends into the nearest and pushes the others far-to-near — half the node fetches and dependent
// steps of a BVH2 walk.  WIDTH = 2 walks the BVH2 in any box format (Node16 / Node32 / Node64: scenes that need f64
// boxes or have no BVH4, and A/B runs).  The box tests only cull and the primitive tests are the exact f64 ones, so
// (prim, t) do not depend on the tree or the format.
// Lane refill: wave p owns the 64-entry chunks p, p + P, p + 2P, ... (P = gridDim.x) of the queue.  When
// RT_REFILL_T or more lanes are idle and chunks are left, the finished lanes store their hits and take the
// next rays.  Launched with P = number of chunks it is the plain one-chunk-per-wave kernel; launched with
// P = resident waves it is persistent.  Every wave exits: chunks are consumed monotonically and the loop
// ends when none is left and no lane holds a ray.  The stack (dynamic LDS, `stack_entries` rows of 64 lanes) holds
// S.stack4 (WIDTH 4) / S.bvh_depth (WIDTH 2) entries per lane (checked by the launcher).
// SHADE: the lane does not store its hit but shades it (shade_path: sky, scatter, Russian roulette) and, if the path
// goes on, traces the scattered ray at once; it takes a new path from the queue only when its own has ended.  One launch
// runs every remaining bounce of a pass: no queue traffic, no launch per bounce, no host in the loop.  Every wave
// still exits: a path's depth grows with every shade and shade_path ends it at max_depth at the latest.
#ifndef RT_REFILL_T
#define RT_REFILL_T 24
#endif
#ifdef RT_TRACE4_WAVES
#define RT_T4_ATTR __attribute__((amdgpu_waves_per_eu(RT_TRACE4_WAVES, RT_TRACE4_WAVES)))
#else
#define RT_T4_ATTR
#endif
struct WalkShade {
    PassParams pp;
    uint32_t depth0;            // depth of the queue's entries
    uint32_t _pad;
    double* path_radiance;
    uint32_t* hist;             // may be null
    uint32_t* stack_ovf;        // OVF kernels: stack rows beyond the LDS ones, [row][launched lane]
    uint32_t ovf_lds_rows;      // OVF kernels: rows kept in LDS (OVF_LDS_ROWS; RT_HIP_OVF_LDS_ROWS lowers it for tests)
    uint32_t _pad2;
};
// OVF: the LDS stack holds OVF_LDS_ROWS entries per lane whatever the tree's worst case is, and deeper entries go to a
// per-lane column in device memory.  A wave's LDS decides how many waves a CU holds: 48 or 64 rows (trees of 5 M+
// triangles) leave 2.5-3 waves per SIMD where the registers allow 4, and these walks are bound by memory latency, i.e. by
// the number of waves in flight.  Real traversals stay far below the worst case the builder computes, so the overflow
// rows are touched by few lanes.
static_assert(OVF_LDS_ROWS == 32, "the OVF kernels' LDS stack");      // (rt_kernels.h; LaunchCtx::ovf_lds_rows lowers it for tests)
// MODE of a walk kernel: plain; OVF (above); ALL = every lane fetches (below, at the loads) — for scenes that live in the
// caches, which are not the ones with deep trees, so the two are never combined.  (A run-time switch instead of an instance
// per mode costs two registers, and the kernel sits exactly at the count that lets a fifth wave of another kernel onto
// the SIMD.)
enum { WALK_PLAIN = 0, WALK_OVF = 1, WALK_ALL = 2 };
template <class BoxT, int WIDTH, bool STATS, bool SHADE, int MODE>
__device__ __forceinline__ void walk_body(const DScene S, const RayQueue q, const QueueGeom qg, double t_min, double t_max,
                                          const HitBuf hb, unsigned long long* counters, const WalkShade sh) {
    // (by VALUE on purpose: through references the compiler turned the base-pointer select of the fetch below into a
    // select of ADDRESSES inside the argument block plus a per-lane load — one more dependent memory round trip per step)
    constexpr bool OVF = MODE == WALK_OVF;
    extern __shared__ uint32_t stack_mem[];
    // The stack pointer is the BYTE ADDRESS of the lane's next free entry in LDS, spa = 256 * entries + 4 * lane: a push is a
    // store at spa, the lane's number rides in its low byte, and neither threadIdx.x nor 4 * threadIdx.x has to stay in a
    // register across the loop (they were the two values the 96-register build of the overflow instance spilled, and reloaded
    // from scratch inside the loop).  An idle lane's spa is 4 * lane: a ray that starts has nothing to reset.
#define RT_LDS_AT(addr) (*reinterpret_cast<uint32_t*>(reinterpret_cast<char*>(stack_mem) + (addr)))
#define RT_ROWS() (spa >> 8)
    // (the overflow address is formed where it is used: a per-lane pointer kept across the loop would cost the kernel the
    // two registers that decide whether a camera-ray or shading wave still fits next to four of its waves on a SIMD)
#define RT_OVF_AT(row) sh.stack_ovf[((uint32_t)(row) * gridDim.x + blockIdx.x) * (uint32_t)TRACE_BLOCK + ((spa & 255u) >> 2)]      // (< 2^32: the launcher checks)
    // `deep` (wave-uniform, set once per step): some lane's stack may reach the overflow rows in this step; otherwise the
    // pushes and pops are the plain LDS ones — one compare and one scalar branch per step instead of a check per access
    const uint32_t lds_rows = sh.ovf_lds_rows;      // (a scalar; only the OVF instances read it)
#define RT_PUSH(v) { if (!OVF || !deep || RT_ROWS() < lds_rows) RT_LDS_AT(spa) = (v); else RT_OVF_AT(RT_ROWS() - lds_rows) = (v); spa += 256u; }
#define RT_POP() (spa -= 256u, (!OVF || !deep || RT_ROWS() < lds_rows) ? RT_LDS_AT(spa) : RT_OVF_AT(RT_ROWS() - lds_rows))
#define RT_ANY() (spa >= 256u)
    // The queue's counts are read once here (size in waves, total) and looked up again per refill through the scalar
    // cache (slot_light): nothing of them stays in a VGPR across the loop — the kernel's register count decides whether
    // a camera-ray or shading wave of another pass still fits next to four of its waves on a SIMD.
    WaveQueue wq;
    wq.load(qg, threadIdx.x);
    if (blockIdx.x == 0) {
        const uint32_t tid = threadIdx.x;
        uint32_t tot = wq.direct ? 0u : wq.cnt;
        for (int off = 32; off; off >>= 1) tot += (uint32_t)__shfl_xor((int)tot, off);
        if (wq.direct) tot = wq.n_direct;
        if (tid == 0) {
            if (SHADE && sh.hist) *sh.hist = tot;          // for the host's launch estimates
            // for the one-shot waves of the k_shade launch that follows on the same queue
            if (!wq.direct) const_cast<uint32_t*>(qg.count)[QUEUE_NWAVES_WORD] = wq.n_waves;
            // stand-alone traversal: every entry is one segment (the trace-and-shade kernel counts per lane instead)
            if (!SHADE && counters && tot) atomicAdd(&counters[STAT_SEGMENTS], (unsigned long long)tot);
        }
    }
    wq.cnt = 0;
    const uint32_t n_chunks = wq.n_waves, stride = gridDim.x;
    uint32_t next_chunk = blockIdx.x, chunk = 0, chunk_pos = 64;      // wave-uniform
    bool busy = false;                                                // this lane holds a ray
    uint32_t slot = 0;
    D3 o = mk(0, 0, 0), d = mk(1, 0, 0);
    D3 thr = mk(1, 1, 1);                                             // SHADE: path throughput, path id, depth of the ray in flight
    uint32_t pid = 0, depth = 0;
    constexpr bool W4 = WIDTH == 4;
    constexpr bool F32_NODES = W4 && std::is_same<BoxT, Box32>::value;      // Node4f (128 B) instead of Node4 (64 B)
    BoxT rb;
    rb.init(S, o, d, t_min, t_max);
    double best = t_max;                                              // (barycentrics go to hb when a hit is accepted:
    uint32_t prim = RT_MISS;                                          //  four registers fewer across the loop)
    uint32_t cur = REF_EMPTY;
    uint32_t spa = 4u * threadIdx.x;
    uint32_t n_nodes = 0, n_prims = 0, n_iters = 0, n_leaves = 0, n_seg = 0;

    for (;;) {
        const bool at_node = !(cur & REF_LEAF);                 // REF_EMPTY has the leaf bit set
        const bool at_leaf = !at_node && cur != REF_EMPTY;
        const unsigned long long m_live = __ballot(at_node || at_leaf);
        bool more = chunk_pos < 64u || next_chunk < n_chunks;
        // SHADE: lanes with something to do in the block below = finished traversals (to shade) + idle lanes while rays are left
        auto shade_service = [&]() -> bool {
            const int n_fin = __popcll(__ballot(busy && cur == REF_EMPTY));
            const int n_idle = more ? __popcll(__ballot(!busy)) : 0;
            return !m_live || n_fin + n_idle >= RT_REFILL_T;
        };
        // (the condition sits in the `if` itself: as a bool variable it cost six VALU instructions per iteration)
        if (SHADE ? shade_service() : (!m_live || (more && __popcll(m_live) <= 64 - RT_REFILL_T))) {
            if (busy && cur == REF_EMPTY) {                           // retire
                if (SHADE) {
                    ++n_seg;
                    D3 no = mk(0, 0, 0), nd = mk(0, 0, 0), nthr = mk(0, 0, 0);
                    const PathIndex ix = decode_path(sh.pp, pid);
                    const bool alive = shade_path(S, sh.pp, depth, pid, ix.pixel, ix.local, slot, o, d, thr, prim, best, hb, sh.path_radiance, no, nd, nthr);
                    if (alive) {                                      // Camera::raycast's recursion (src/camera.rs:303): the same lane goes on
                        o = no; d = nd; thr = nthr; ++depth;
                        rb.init(S, o, d, t_min, t_max);
                        best = t_max; prim = RT_MISS;
                        cur = S.root;      // (its stack is empty: spa is back at 4 * lane)
                    } else {
                        busy = false;
                    }
                } else {
                    hb.t[slot] = prim == RT_MISS ? 0.0 : best;
                    hb.prim[slot] = prim;
                    busy = false;
                }
            }
            while (more) {                                            // refill
                const unsigned long long m_idle = __ballot(!busy);
                if (!m_idle) break;
                if (chunk_pos >= 64u) {
                    chunk = next_chunk; next_chunk += stride; chunk_pos = 0;
                }
                const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(m_idle >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m_idle, 0u));
                const uint32_t take = min((uint32_t)__popcll(m_idle), 64u - chunk_pos);
                uint32_t i = 0;
                if (!busy && rank < take && wq.slot_light(qg, chunk, chunk_pos + rank, i)) {
                    slot = i; busy = true;
                    o = mk(q.o[0][i], q.o[1][i], q.o[2][i]);
                    d = mk(q.d[0][i], q.d[1][i], q.d[2][i]);
                    if (SHADE) {
                        thr = mk(q.thr[0][i], q.thr[1][i], q.thr[2][i]);
                        pid = q.pid[i];
                        depth = sh.depth0;
                    }
                    rb.init(S, o, d, t_min, t_max);
                    best = t_max; prim = RT_MISS;
                    cur = S.root;      // (its stack is empty: spa is back at 4 * lane)
                }
                chunk_pos += take;
                more = chunk_pos < 64u || next_chunk < n_chunks;
            }
            if (!__ballot(busy)) break;
            continue;
        }
        if (STATS) ++n_iters;
        const bool deep = OVF && __ballot(RT_ROWS() + 3u > lds_rows) != 0ull;
        // One dependent memory round trip per iteration for every lane: a node lane fetches its 64-byte node,
        // a leaf lane ONE primitive of its leaf (80-byte triangle / 64-byte sphere); a leaf of n primitives
        // takes n iterations (the reference keeps count and first, so it is advanced in place).  All loads are
        // issued before anything waits, whatever mix of node and leaf lanes the wave holds.  (The BVH2 walk loads its
        // node inside the node step instead: it is the fallback, not the fast path.)
        const bool tri_leaf = at_leaf && !(cur & REF_SPHERE);
        // BVH4 walks of scenes that live in the caches (MODE ALL; DScene::all_lanes_load, set at scene creation): EVERY lane loads — a lane
        // without work reads node 0, one line for all of them — so the loads need no masked region and the fetch registers no
        // zeroing: 30 instructions fewer per step (+1.2 % frame rate on the 1 M-triangle scene).  Scenes beyond the caches are
        // bound by memory requests, not by instruction issue: there the extra requests cost 1 %, and the lanes sit out.
        // (Every scene buffer is padded by 256 bytes: the widest record read here is 112 bytes, whatever lies at the address.)
        constexpr bool all_load = W4 && MODE == WALK_ALL;
        const bool idle = all_load && cur == REF_EMPTY;
        const uint32_t first = idle ? 0u : cur & REF_FIRST_MASK;
        const bool node_load = W4 && (at_node || idle);
        const char* src;
        if constexpr (W4) {
            // BVH4 references carry the record's offset in the pool in 16-byte units (DScene::pool4): one shift, and the
            // 32-bit byte offset rides on the scalar base — no select between three arrays and strides
            // (the four flag bits of a reference fall off the top of the shift; a lane without work fetches record 0)
            src = S.pool4 + (uint64_t)(idle ? 0u : cur << 4);
        } else {
            const char* base_ = tri_leaf ? reinterpret_cast<const char*>(S.tris) : reinterpret_cast<const char*>(S.sphs);
            const uint32_t stride_ = tri_leaf ? (uint32_t)sizeof(TriGeom) : (uint32_t)sizeof(SphGeom);
            src = base_ + (uint64_t)first * stride_;
        }
        const uint4* p = reinterpret_cast<const uint4*>(src);
        uint4 a = make_uint4(0, 0, 0, 0), b = a, c = a, r = a, e = a, f = a, g = a;
        if (all_load) {
            a = p[0]; b = p[1]; c = p[2]; r = p[3]; e = p[4];
            if (F32_NODES) { f = p[5]; g = p[6]; }
        } else {
            if (node_load || at_leaf) { a = p[0]; b = p[1]; c = p[2]; r = p[3]; }      // idle lanes hold REF_EMPTY: no address
            if (tri_leaf || (F32_NODES && at_node)) e = p[4];
            if (F32_NODES && at_node) { f = p[5]; g = p[6]; }
        }
        if (at_node) {
            if (STATS) ++n_nodes;
            if constexpr (W4) {
                const uint32_t w[24] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w, c.x, c.y, c.z, c.w,
                                        r.x, r.y, r.z, r.w, e.x, e.y, e.z, e.w, f.x, f.y, f.z, f.w};
                uint32_t ref[4] = {r.x, r.y, r.z, r.w};
                if (F32_NODES) { ref[0] = g.x; ref[1] = g.y; ref[2] = g.z; ref[3] = g.w; }
                float key[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    float tn = 0.f;
                    bool h;
                    if constexpr (F32_NODES) {          // child k owns floats 6k .. 6k+5
                        const bool neg[3] = {rb.ix < 0.f, rb.iy < 0.f, rb.iz < 0.f};
                        float near[3], far[3];
                        for (int j = 0; j < 3; ++j) {
                            const float lo = __uint_as_float(w[6 * k + j]), hi = __uint_as_float(w[6 * k + 3 + j]);
                            near[j] = neg[j] ? hi : lo; far[j] = neg[j] ? lo : hi;
                        }
                        const bool hh = rb.test_picked(near, far, tn);
                        h = hh & (ref[k] != REF_EMPTY);
                    } else if constexpr (std::is_same<BoxT, Box16>::value) {
                        // u16 number j of the node sits in word j/2, half j%2; child k owns numbers 6k .. 6k+5.
                        // all four children are tested without a branch on the empty ones (an empty slot's box is all zeros):
                        // four masked regions per node step cost more than the arithmetic they skip
                        const bool hh = rb.test_child(w[3 * k], w[3 * k + 1], w[3 * k + 2], tn);
                        h = hh & (ref[k] != REF_EMPTY);
                    } else {
                        h = false;                      // (no BVH4 with f64 boxes exists: never instantiated at run time)
                    }
                    key[k] = h ? tn : INFINITY;                 // misses carry no child: wherever the sort leaves them (a NaN entry
                    ref[k] = h ? ref[k] : REF_EMPTY;            // distance compares false both ways), they are skipped below
                }
#define RT_CE(x, y) { const bool sw = key[y] < key[x]; const float kx = sw ? key[y] : key[x], ky = sw ? key[x] : key[y]; \
                      const uint32_t rx = sw ? ref[y] : ref[x], ry = sw ? ref[x] : ref[y]; key[x] = kx; key[y] = ky; ref[x] = rx; ref[y] = ry; }
                RT_CE(0, 1) RT_CE(2, 3) RT_CE(0, 2) RT_CE(1, 3) RT_CE(1, 2)
#undef RT_CE
                if (ref[3] != REF_EMPTY) RT_PUSH(ref[3])
                if (ref[2] != REF_EMPTY) RT_PUSH(ref[2])
                if (ref[1] != REF_EMPTY) RT_PUSH(ref[1])
                cur = ref[0] != REF_EMPTY ? ref[0] : (RT_ANY() ? RT_POP() : REF_EMPTY);
            } else {
                typename BoxT::Scalar lo_l[3], hi_l[3], lo_r[3], hi_r[3], tl = 0, tr = 0;
                uint32_t cl, cr;
                rb.fetch(S, cur, lo_l, hi_l, lo_r, hi_r, cl, cr);
                const bool hl = rb.test(lo_l, hi_l, tl);
                const bool hr = cr != REF_EMPTY && rb.test(lo_r, hi_r, tr);
                if (hl && hr) {
                    const bool left_first = tl <= tr;
                    RT_PUSH(left_first ? cr : cl)
                    cur = left_first ? cl : cr;
                } else if (hl) {
                    cur = cl;
                } else if (hr) {
                    cur = cr;
                } else {
                    cur = RT_ANY() ? RT_POP() : REF_EMPTY;
                }
            }
        }
        if (at_leaf) {
            if (STATS) { ++n_prims; ++n_leaves; }
            if (tri_leaf) {                                           // TriGeom: a, e1, e2 (9 doubles), prim
                const D3 ta = mk(mkd(a.x, a.y), mkd(a.z, a.w), mkd(b.x, b.y));
                const D3 e1 = mk(mkd(b.z, b.w), mkd(c.x, c.y), mkd(c.z, c.w));
                const D3 e2 = mk(mkd(r.x, r.y), mkd(r.z, r.w), mkd(e.x, e.y));
                // (acceptance inside the test's success path — tri_test_do: nothing of a failed test lives on)
                tri_test_do(ta, e1, e2, o, d, t_min, best, [&](double t, double u, double v) {
                    best = t; prim = e.z; rb.nearest(best);
                    hb.bu[slot] = u; hb.bv[slot] = v;
                });
            } else {                                                  // SphGeom: c, r, rot2 (7 doubles), prim
                const D3 sc = mk(mkd(a.x, a.y), mkd(a.z, a.w), mkd(b.x, b.y));
                const double sr = mkd(b.z, b.w);
                const D3 rot2 = mk(mkd(c.x, c.y), mkd(c.z, c.w), mkd(r.x, r.y));
                sph_test_do(sc, sr, rot2, o, d, t_min, best, [&](double t) { best = t; prim = r.z; rb.nearest(best); });
            }
            // next primitive of the same leaf, or pop
            // (BVH4: references count 16-byte units — a triangle is 5 of them, a sphere 4)
            const uint32_t next_ = W4 ? ((cur & REF_SPHERE) ? (uint32_t)(sizeof(SphGeom) / 16) : (uint32_t)(sizeof(TriGeom) / 16)) : 1u;
            cur = (cur & 0x30000000u) ? cur - 0x10000000u + next_ : (RT_ANY() ? RT_POP() : REF_EMPTY);
        }
    }
#undef RT_PUSH
#undef RT_POP
#undef RT_ANY
#undef RT_ROWS
#undef RT_LDS_AT
#undef RT_OVF_AT
    // (one wave per workgroup: the lane's number, without keeping threadIdx.x alive across the loop)
    const uint32_t tid = __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
    if (SHADE && counters) {
        unsigned long long sg = n_seg;
        for (int off = 32; off; off >>= 1) sg += __shfl_xor(sg, off);
        if (tid == 0 && sg) atomicAdd(&counters[STAT_SEGMENTS], sg);
    }
    if (STATS) {
        unsigned long long a = n_nodes, b = n_prims, c = n_leaves;
        for (int off = 32; off; off >>= 1) {
            a += __shfl_xor(a, off);
            b += __shfl_xor(b, off);
            c += __shfl_xor(c, off);
        }
        if (tid == 0) {
            atomicAdd(&counters[STAT_NODES], a);
            atomicAdd(&counters[STAT_PRIMS], b);
            atomicAdd(&counters[STAT_ITERS], (unsigned long long)n_iters);
            atomicAdd(&counters[STAT_LEAVES], c);
        }
    }
}

// The two kernels over walk_body: k_walk stores hits (stand-alone traversal: first bounce, rt_trace_closest, ...),
// k_walk_shade is the persistent trace-and-shade kernel.  Separate __global__ functions so that each has its own
// register budget: RT_TAIL_WAVES caps k_walk_shade's (its shading code wants ~200 VGPRs; held by long-lived waves they
// would keep the other passes' kernels out of the SIMDs).
template <class BoxT, int WIDTH, bool STATS, int MODE>
__global__ RT_T4_ATTR __launch_bounds__(TRACE_BLOCK, RT_TRACE_WAVES) void k_walk(DScene S, RayQueue q, QueueGeom qg, double t_min, double t_max,
                                                                       HitBuf hb, unsigned long long* counters, WalkShade sh) {
    walk_body<BoxT, WIDTH, STATS, false, MODE>(S, q, qg, t_min, t_max, hb, counters, sh);
}
// k_walk5: the same kernel held to 96 VGPRs.  The BVH4 / Box16 instances fit into 96 registers without a spill (the
// overflow-stack instance only since the stack pointer carries the lane number: before, it spilled threadIdx.x and
// 4 * threadIdx.x and reloaded them from scratch inside the loop, where a reload waits for every load in flight: -20 % of a
// frame).  With several passes in flight it is worth 1.5-2 % of a frame on the 1 M-triangle scene, 3.6 % on the 5 M and 1.9 %
// on the 20 M scene: its waves leave more of the register file to the other passes' kernels.  Alone (one pass in flight,
// rt_trace_closest) it is 8 % SLOWER than k_walk — the 96-register schedule is the worse code, and a fifth resident wave
// does not make up for it — so the renderer asks for it only when several passes are in flight (LaunchCtx::five).
template <class BoxT, int WIDTH, bool STATS, int MODE>
__global__ __attribute__((amdgpu_waves_per_eu(5, 5))) __launch_bounds__(TRACE_BLOCK) void k_walk5(DScene S, RayQueue q, QueueGeom qg, double t_min,
                                                                                                    double t_max, HitBuf hb, unsigned long long* counters,
                                                                                                    WalkShade sh) {
    walk_body<BoxT, WIDTH, STATS, false, MODE>(S, q, qg, t_min, t_max, hb, counters, sh);
}
#ifdef RT_TAIL_WAVES
#define RT_TAIL_ATTR __attribute__((amdgpu_waves_per_eu(RT_TAIL_WAVES, RT_TAIL_WAVES)))
#else
#define RT_TAIL_ATTR
#endif
template <class BoxT, int WIDTH, bool STATS, int MODE>
__global__ RT_TAIL_ATTR __launch_bounds__(TRACE_BLOCK, RT_TRACE_WAVES) void k_walk_shade(DScene S, RayQueue q, QueueGeom qg, double t_min, double t_max,
                                                                           HitBuf hb, unsigned long long* counters, WalkShade sh) {
    walk_body<BoxT, WIDTH, STATS, true, MODE>(S, q, qg, t_min, t_max, hb, counters, sh);
}

// ---------------------------------------------------------------------------------------------
// K1w — per-lane traversal of the BVH8 (rt_layout.h: Node8): an alternative to the BVH4 walk (RT_HIP_NODE_BOXES=w8).
// ---------------------------------------------------------------------------------------------
// Ray-side state.  A plane of a child box is the real number  origin_k + q * 2^e_k  (q = 0..255), so its ray parameter is
//     t = (origin_k + q 2^e_k - o_k) / d_k = q * (2^e_k / d_k) + (origin_k - o_k) / d_k.
// Per ray (Box32::set_axis):  idf = fl32(1/d),  K = fl32(o * idf),  c = |K| 2^-21,  KN = fl32(-K - c),  KF = fl32(-K + c).
// Per node and axis:          adj = idf * 2^e (exact: a power of two),  s = |adj| 2^-15,
//                             BN = fl32(fl32(origin * idf + KN) - s),   BF = fl32(fl32(origin * idf + KF) + s)   (two fma each)
// Per plane:                  Y = fl32(q * adj + BN)  for near planes,  fl32(q * adj + BF)  for far planes        (one fma)
// Error budget against the exact t, u = 2^-24:  q * adj + origin * idf = (origin + q 2^e) * idf exactly, and
//     (origin + q 2^e) * idf - K = (1 + e2)[t - A e3],  A = o / d,  |e2|, |e3| <= u                 (Box32's budget: roundings of idf and K)
// The roundings on top are KN's (<= u (|K| + c)), BN's two (each <= u |BN|, |BN| <= |Y| + 255 |adj| + s) and Y's (<= u |Y|):
//     t - Y >= -(4.2 u) |Y| + (c - 3.2 u |K|) + (s - 2.1 u 255 |adj|)  >=  -2^-22 |Y|       since c = 8 u |K| and s = 512 u |adj|
// and symmetrically for far planes: after  Y -> Y -/+ |Y| 2^-22  (applied once, after max3 / min3: it is monotone) a near
// value is never above, a far value never below the exact one.  NaNs (0 * inf, inf - inf: rays parallel to a slab, products
// beyond the f32 range) drop out of fmaxf / fminf or reach the negated comparisons, which accept; an infinity that survives has
// the sign of the exact value (every term that can overflow carries the sign of its exact counterpart, and two terms of
// opposite sign that both overflow give NaN).  The test only culls: primitives are intersected in f64 as in the other walks.
struct Box8 : Box32 {
    uint32_t oinv;     // bit k set: the ray runs in + direction on axis k (= 7 - octant)
    uint32_t psel;     // v_perm_b32 selector moving byte j to byte j ^ (oinv >> 1): the byte part of the priority permutation
    __device__ __forceinline__ void init(const DScene& S, D3 o, D3 d, double t_min, double t_max) {
        Box32::init(S, o, d, t_min, t_max);
        oinv = (ix < 0.f ? 0u : 1u) | (iy < 0.f ? 0u : 2u) | (iz < 0.f ? 0u : 4u);
        psel = 0x03020100u ^ ((oinv >> 1) * 0x01010101u);
    }
    // nibble s -> nibble s ^ oinv: slot order to priority order (the highest nibble is the child to visit first)
    __device__ __forceinline__ uint32_t to_priority(uint32_t x) const {
        const uint32_t sw = ((x & 0x0F0F0F0Fu) << 4) | ((x >> 4) & 0x0F0F0F0Fu);
        x = (oinv & 1u) ? sw : x;
        return __builtin_amdgcn_perm(x, x, psel);
    }
};
constexpr uint32_t G_HITS = 0x88888888u;      // pending children of a node group: top bit of the nibble at priority p
constexpr uint32_t G_IMASK = 0x11111111u;     // internal-child slots of the group's node: low bit of the nibble at slot s

// One step = one dependent memory round trip per lane, as in walk_body: a lane fetches ONE 80-byte record — the next
// node of its pending group, or the next primitive of the node it tested last.  Per-lane state: the group (base =
// first record of the node's block, g = pending internal children by priority + the node's internal-slot mask) and the
// node's pending primitives (tb = hit bits by slot nibble, tp = the node's pmask, for the rank of a bit among the
// records).  The stack holds groups only (two words each), at most one per level of internal nodes: S.depth8 entries.
// A node's primitives are tested before its children are entered; children are entered in octant order.  Lane refill,
// retiring, shading (SHADE) and the statistics are those of walk_body.
template <bool STATS, bool SHADE>
__device__ __forceinline__ void walk8_body(const DScene S, const RayQueue q, const QueueGeom qg, double t_min, double t_max,
                                           const HitBuf hb, unsigned long long* counters, const WalkShade sh) {
    extern __shared__ uint32_t stack_mem[];
    uint32_t (*stack)[TRACE_BLOCK] = reinterpret_cast<uint32_t (*)[TRACE_BLOCK]>(stack_mem);
    const uint32_t tid = threadIdx.x;
    WaveQueue wq;
    wq.load(qg, tid);
    if (blockIdx.x == 0) {
        uint32_t tot = wq.direct ? 0u : wq.cnt;
        for (int off = 32; off; off >>= 1) tot += (uint32_t)__shfl_xor((int)tot, off);
        if (wq.direct) tot = wq.n_direct;
        if (tid == 0) {
            if (SHADE && sh.hist) *sh.hist = tot;
            if (!wq.direct) const_cast<uint32_t*>(qg.count)[QUEUE_NWAVES_WORD] = wq.n_waves;
            if (!SHADE && counters && tot) atomicAdd(&counters[STAT_SEGMENTS], (unsigned long long)tot);
        }
    }
    wq.cnt = 0;
    const uint32_t n_chunks = wq.n_waves, stride = gridDim.x;
    uint32_t next_chunk = blockIdx.x, chunk = 0, chunk_pos = 64;      // wave-uniform
    bool busy = false;
    uint32_t slot = 0;
    D3 o = mk(0, 0, 0), d = mk(1, 0, 0);
    D3 thr = mk(1, 1, 1);
    uint32_t pid = 0, depth = 0;
    Box8 rb;
    rb.init(S, o, d, t_min, t_max);
    double best = t_max;
    uint32_t prim = RT_MISS;
    uint32_t gbase = 0, g = 0, tb = 0, tp = 0;
    int sp = 0;
    uint32_t n_nodes = 0, n_prims = 0, n_iters = 0, n_seg = 0;
    // a ray starts with the root (record 0) as the only child of a virtual group: slot 0, pending at priority 0 ^ oinv
#define RT_START_RAY() { gbase = 0u; g = S.root == REF_EMPTY ? 0u : ((0x8u << (4u * rb.oinv)) | 0x1u); tb = 0u; tp = 0u; sp = 0; best = t_max; prim = RT_MISS; }

    for (;;) {
        const bool at_prim = tb != 0u;
        const bool at_node = !at_prim && (g & G_HITS) != 0u;
        const unsigned long long m_live = __ballot(at_prim || at_node);
        bool more = chunk_pos < 64u || next_chunk < n_chunks;
        auto shade_service = [&]() -> bool {
            const int n_fin = __popcll(__ballot(busy && !at_prim && !at_node));
            const int n_idle = more ? __popcll(__ballot(!busy)) : 0;
            return !m_live || n_fin + n_idle >= RT_REFILL_T;
        };
        if (SHADE ? shade_service() : (!m_live || (more && __popcll(m_live) <= 64 - RT_REFILL_T))) {
            if (busy && !at_prim && !at_node) {                       // retire
                if (SHADE) {
                    ++n_seg;
                    D3 no = mk(0, 0, 0), nd = mk(0, 0, 0), nthr = mk(0, 0, 0);
                    const PathIndex ix = decode_path(sh.pp, pid);
                    const bool alive = shade_path(S, sh.pp, depth, pid, ix.pixel, ix.local, slot, o, d, thr, prim, best, hb, sh.path_radiance, no, nd, nthr);
                    if (alive) {
                        o = no; d = nd; thr = nthr; ++depth;
                        rb.init(S, o, d, t_min, t_max);
                        RT_START_RAY()
                    } else {
                        busy = false;
                    }
                } else {
                    hb.t[slot] = prim == RT_MISS ? 0.0 : best;
                    hb.prim[slot] = prim;
                    busy = false;
                }
            }
            while (more) {                                            // refill
                const unsigned long long m_idle = __ballot(!busy);
                if (!m_idle) break;
                if (chunk_pos >= 64u) {
                    chunk = next_chunk; next_chunk += stride; chunk_pos = 0;
                }
                const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(m_idle >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m_idle, 0u));
                const uint32_t take = min((uint32_t)__popcll(m_idle), 64u - chunk_pos);
                uint32_t i = 0;
                if (!busy && rank < take && wq.slot_light(qg, chunk, chunk_pos + rank, i)) {
                    slot = i; busy = true;
                    o = mk(q.o[0][i], q.o[1][i], q.o[2][i]);
                    d = mk(q.d[0][i], q.d[1][i], q.d[2][i]);
                    if (SHADE) {
                        thr = mk(q.thr[0][i], q.thr[1][i], q.thr[2][i]);
                        pid = q.pid[i];
                        depth = sh.depth0;
                    }
                    rb.init(S, o, d, t_min, t_max);
                    RT_START_RAY()
                }
                chunk_pos += take;
                more = chunk_pos < 64u || next_chunk < n_chunks;
            }
            if (!__ballot(busy)) break;
            continue;
        }
        if (STATS) ++n_iters;
        // ---- which record: the next pending primitive of the node tested last, else the next pending child
        uint32_t rec = 0;
        if (at_prim) {
            const uint32_t below = ~tb & (tb - 1u);                               // the bits below the lowest pending one
            rec = gbase + (uint32_t)__popc(g & G_IMASK) + (uint32_t)__popc(tp & below);      // behind the internal children, by rank
            tb &= tb - 1u;
        } else if (at_node) {
            const uint32_t b = 31u - (uint32_t)__clz((int)(g & G_HITS));          // highest pending priority
            g &= ~(1u << b);
            const uint32_t s_ = (b >> 2) ^ rb.oinv;                               // its slot
            rec = gbase + (uint32_t)__popc(g & G_IMASK & ((1u << (4u * s_)) - 1u));
            if (g & G_HITS) {                                                     // the rest of the group waits on the stack
                stack[2 * sp][tid] = gbase;
                stack[2 * sp + 1][tid] = g;
                ++sp;
            }
        }
        const uint4* p = reinterpret_cast<const uint4*>(S.pool8 + (uint64_t)rec * PRIM_RECORD_BYTES);
        uint4 a = make_uint4(0, 0, 0, 0), b = a, c = a, r = a, e = a;
        if (at_prim || at_node) { a = p[0]; b = p[1]; c = p[2]; r = p[3]; e = p[4]; }
        if (at_node) {
            if (STATS) ++n_nodes;
            // Node8: a = origin, ex | b = base, inib, pmask | c = qlo_x, qlo_y | r = qlo_z, qhi_x | e = qhi_y, qhi_z (8 bytes each)
            const float adjx = rb.ix * __uint_as_float((a.w & 0xFFu) << 23);
            const float adjy = rb.iy * __uint_as_float(((a.w >> 8) & 0xFFu) << 23);
            const float adjz = rb.iz * __uint_as_float(((a.w >> 16) & 0xFFu) << 23);
            const float s15 = 0x1.0p-15f, r22 = 0x1.0p-22f;
            const float ox = __uint_as_float(a.x), oy = __uint_as_float(a.y), oz = __uint_as_float(a.z);
            const float bnx = fmaf(-fabsf(adjx), s15, fmaf(ox, rb.ix, rb.nx)), bfx = fmaf(fabsf(adjx), s15, fmaf(ox, rb.ix, rb.fx));
            const float bny = fmaf(-fabsf(adjy), s15, fmaf(oy, rb.iy, rb.ny)), bfy = fmaf(fabsf(adjy), s15, fmaf(oy, rb.iy, rb.fy));
            const float bnz = fmaf(-fabsf(adjz), s15, fmaf(oz, rb.iz, rb.nz)), bfz = fmaf(fabsf(adjz), s15, fmaf(oz, rb.iz, rb.fz));
            // near and far planes of each axis by the sign of the direction: whole words, eight children at a time
            const bool ngx = rb.ix < 0.f, ngy = rb.iy < 0.f, ngz = rb.iz < 0.f;
            const uint32_t nxw[2] = {ngx ? r.z : c.x, ngx ? r.w : c.y}, fxw[2] = {ngx ? c.x : r.z, ngx ? c.y : r.w};
            const uint32_t nyw[2] = {ngy ? e.x : c.z, ngy ? e.y : c.w}, fyw[2] = {ngy ? c.z : e.x, ngy ? c.w : e.y};
            const uint32_t nzw[2] = {ngz ? e.z : r.x, ngz ? e.w : r.y}, fzw[2] = {ngz ? r.x : e.z, ngz ? r.y : e.w};
            uint32_t hitnib = 0u;
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const int w = k >> 2, shf = 8 * (k & 3);
                const float qnx = (float)((nxw[w] >> shf) & 0xFFu), qfx = (float)((fxw[w] >> shf) & 0xFFu);
                const float qny = (float)((nyw[w] >> shf) & 0xFFu), qfy = (float)((fyw[w] >> shf) & 0xFFu);
                const float qnz = (float)((nzw[w] >> shf) & 0xFFu), qfz = (float)((fzw[w] >> shf) & 0xFFu);
                float tn = fmaxf(fmaxf(fmaf(qnx, adjx, bnx), fmaf(qny, adjy, bny)), fmaf(qnz, adjz, bnz));
                float tf = fminf(fminf(fmaf(qfx, adjx, bfx), fmaf(qfy, adjy, bfy)), fmaf(qfz, adjz, bfz));
                tn = fmaf(-fabsf(tn), r22, tn);
                tf = fmaf(fabsf(tf), r22, tf);
                const bool h = !(tn > tf) && !(tf < rb.lo_t) && !(tn > rb.hi_t);
                hitnib |= h ? (0xFu << (4 * k)) : 0u;
            }
            // (an empty slot has neither an inib nor a pmask nibble: whatever its test said, it drops out here)
            gbase = b.x;
            g = (rb.to_priority(hitnib & b.y) & G_HITS) | (b.y & G_IMASK);
            tb = hitnib & b.z;
            tp = b.z;
        }
        if (at_prim) {
            if (STATS) ++n_prims;
            if (e.w == (uint32_t)PRIM_KIND_TRIANGLE) {                // TriGeom: a, e1, e2 (9 doubles), prim, kind
                const D3 ta = mk(mkd(a.x, a.y), mkd(a.z, a.w), mkd(b.x, b.y));
                const D3 e1 = mk(mkd(b.z, b.w), mkd(c.x, c.y), mkd(c.z, c.w));
                const D3 e2 = mk(mkd(r.x, r.y), mkd(r.z, r.w), mkd(e.x, e.y));
                tri_test_do(ta, e1, e2, o, d, t_min, best, [&](double t, double u, double v) {
                    best = t; prim = e.z; rb.nearest(best);
                    hb.bu[slot] = u; hb.bv[slot] = v;
                });
            } else {                                                  // SphGeom: c, r, rot2 (7 doubles), prim
                const D3 sc = mk(mkd(a.x, a.y), mkd(a.z, a.w), mkd(b.x, b.y));
                const double sr = mkd(b.z, b.w);
                const D3 rot2 = mk(mkd(c.x, c.y), mkd(c.z, c.w), mkd(r.x, r.y));
                sph_test_do(sc, sr, rot2, o, d, t_min, best, [&](double t) { best = t; prim = r.z; rb.nearest(best); });
            }
        }
        // nothing pending at this node any more: back to the group that was set aside last
        if ((at_prim || at_node) && tb == 0u && !(g & G_HITS) && sp) {
            --sp;
            gbase = stack[2 * sp][tid];
            g = stack[2 * sp + 1][tid];
        }
    }
#undef RT_START_RAY
    if (SHADE && counters) {
        unsigned long long sg = n_seg;
        for (int off = 32; off; off >>= 1) sg += __shfl_xor(sg, off);
        if (tid == 0 && sg) atomicAdd(&counters[STAT_SEGMENTS], sg);
    }
    if (STATS) {
        unsigned long long a = n_nodes, b = n_prims;
        for (int off = 32; off; off >>= 1) {
            a += __shfl_xor(a, off);
            b += __shfl_xor(b, off);
        }
        if (tid == 0) {
            atomicAdd(&counters[STAT_NODES], a);
            atomicAdd(&counters[STAT_PRIMS], b);
            atomicAdd(&counters[STAT_ITERS], (unsigned long long)n_iters);
            atomicAdd(&counters[STAT_LEAVES], b);
        }
    }
}
template <bool STATS>
__global__ __launch_bounds__(TRACE_BLOCK, RT_TRACE_WAVES) void k_walk8(DScene S, RayQueue q, QueueGeom qg, double t_min, double t_max, HitBuf hb,
                                                                      unsigned long long* counters, WalkShade sh) {
    walk8_body<STATS, false>(S, q, qg, t_min, t_max, hb, counters, sh);
}
template <bool STATS>
__global__ RT_TAIL_ATTR __launch_bounds__(TRACE_BLOCK, RT_TRACE_WAVES) void k_walk8_shade(DScene S, RayQueue q, QueueGeom qg, double t_min, double t_max,
                                                                                      HitBuf hb, unsigned long long* counters, WalkShade sh) {
    walk8_body<STATS, true>(S, q, qg, t_min, t_max, hb, counters, sh);
}

// ---------------------------------------------------------------------------------------------
// K1p — closest hit for COHERENT waves (camera rays of an 8x8 pixel tile): packet traversal.
// The wave walks the BVH once: the current node, the stack and the loop are wave-uniform, nodes
// and primitives are fetched through the SCALAR cache (one s_load per node for all 64 rays instead
// of 4 vector loads per lane — k_trace is bound by the vector-memory address unit, TA busy 74-93 %),
// and every lane still runs its own conservative box test and its own exact f64 primitive tests
// with its own nearest-t, so each ray's (prim, t) is what the per-ray traversal finds (ties
// excepted).  A 64-bit lane mask travels with every node reference: lanes whose ray missed the
// box of a subtree skip it.
// ---------------------------------------------------------------------------------------------
typedef uint32_t u32x16 __attribute__((ext_vector_type(16)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
#define RT_AS4 __attribute__((address_space(4)))

// p must be wave-uniform: loads through the constant address space become s_load_dwordx16/x4
__device__ __forceinline__ u32x16 uload16(const void* p) { return *(const RT_AS4 u32x16*)(uintptr_t)p; }
__device__ __forceinline__ u32x4 uload4(const void* p) { return *(const RT_AS4 u32x4*)(uintptr_t)p; }

struct PacketEntry { uint32_t ref, mask_lo, mask_hi; };


// GEN: the wave generates its own camera rays (K0 fused in: saves the queue round trip of k_raygen)
// and stores them for k_shade.
// FUSE (with GEN): the wave also shades its hits (depth 0) and appends the survivors to qo, so neither the
// camera rays nor their hit records travel through HBM (k_shade at depth 0 is bound by exactly that traffic).
// Six waves per SIMD (80 VGPRs, nine spilled) instead of the 85 registers / five waves the compiler picks on its own.
// Frame rate with 5 / 6 / 7 / 8 waves: 1 M-triangle scene 6125 / 6227 / 6310 / 6150, cover scene 6450 / 6487 / 6340 / - Msamples/s.
#ifndef RT_PK_WAVES
#define RT_PK_WAVES 6
#endif
#define RT_PK_ATTR __attribute__((amdgpu_waves_per_eu(RT_PK_WAVES, RT_PK_WAVES)))
template <bool STATS, bool GEN, bool FUSE>
__global__ RT_PK_ATTR __launch_bounds__(TRACE_BLOCK) void k_trace_packet(DScene S, PassParams pp, RayQueue q, QueueGeom qg, double t_min,
                                                               double t_max, HitBuf hb, unsigned long long* counters,
                                                               RayQueue qo, uint32_t* out_counts, double* path_radiance) {
    __shared__ PacketEntry stack[BVH_MAX_DEPTH];
    const uint32_t lane = threadIdx.x;
    uint32_t i = 0;
    const bool active = direct_slot(qg, blockIdx.x * TRACE_BLOCK + lane, i);      // camera rays: always a dense queue
    D3 o = mk(0, 0, 0), d = mk(1, 0, 0);
    PathIndex ix = {0u, 0u, 0u, 0u};          // decoded once: the camera ray and the depth-0 shading share it
    if (active) {
        if (GEN) {
            const uint32_t pid = blockIdx.x * TRACE_BLOCK + lane;
            ix = decode_path(pp, pid);
            camera_ray(pp, ix, o, d);
            if (!FUSE) {
                q.o[0][i] = o.x; q.o[1][i] = o.y; q.o[2][i] = o.z;
                q.d[0][i] = d.x; q.d[1][i] = d.y; q.d[2][i] = d.z;
                q.pid[i] = pid;
            }
        } else {
            o = mk(q.o[0][i], q.o[1][i], q.o[2][i]);
            d = mk(q.d[0][i], q.d[1][i], q.d[2][i]);
        }
    }
    Box32 rb;
    rb.init(S, o, d, t_min, t_max);
    double best = t_max, bu = 0.0, bv = 0.0;
    uint32_t prim = RT_MISS;
    uint32_t n_nodes = 0, n_prims = 0, w_nodes = 0, w_prims = 0;      // per lane / per wave (records the wave fetched)
    uint32_t w_dead = 0;                                              // node records no lane's ray entered a child of
    unsigned long long mask = __ballot(active);
    // Camera rays of a few neighbouring pixels nearly always agree on the sign of each direction component; then the
    // near and far plane of every box are the same for the whole wave and are picked with scalar selects.
    const unsigned long long neg_x = __ballot(active && rb.ix < 0.f), neg_y = __ballot(active && rb.iy < 0.f),
                             neg_z = __ballot(active && rb.iz < 0.f);
    const bool same_signs = (neg_x == 0 || neg_x == mask) && (neg_y == 0 || neg_y == mask) && (neg_z == 0 || neg_z == mask);
    uint32_t cur = (mask && S.root != REF_EMPTY) ? S.root : REF_EMPTY;
    int sp = 0;

    while (cur != REF_EMPTY) {
        if (!(cur & REF_LEAF)) {
            const u32x16 nd = uload16(&S.nodes32[cur]);
            const uint32_t cl = nd[12], cr = nd[13];
            const float lo_l[3] = {__uint_as_float(nd[0]), __uint_as_float(nd[1]), __uint_as_float(nd[2])};
            const float hi_l[3] = {__uint_as_float(nd[3]), __uint_as_float(nd[4]), __uint_as_float(nd[5])};
            const float lo_r[3] = {__uint_as_float(nd[6]), __uint_as_float(nd[7]), __uint_as_float(nd[8])};
            const float hi_r[3] = {__uint_as_float(nd[9]), __uint_as_float(nd[10]), __uint_as_float(nd[11])};
            float tl = 0.f, tr = 0.f, tfl = 0.f, tfr = 0.f;      // entry and exit parameter of the two boxes
            // both boxes tested without masked regions (the node is wave-uniform, the arithmetic is per lane)
            if (same_signs) {
                // four scalar selects per axis behind one compare of the axis' lane mask (the planes sit in SGPRs; written
                // out because the compiler sets the condition code again before every select)
                uint32_t n_l[3], f_l[3], n_r[3], f_r[3];
#define RT_PICK(j, flag) asm("s_cmp_lg_u64 %4, 0\n\ts_cselect_b32 %0, %6, %5\n\ts_cselect_b32 %1, %5, %6\n\t" \
                             "s_cselect_b32 %2, %8, %7\n\ts_cselect_b32 %3, %7, %8" \
                             : "=&s"(n_l[j]), "=&s"(f_l[j]), "=&s"(n_r[j]), "=&s"(f_r[j]) \
                             : "s"(flag), "s"(nd[j]), "s"(nd[3 + j]), "s"(nd[6 + j]), "s"(nd[9 + j]) : "scc")
                RT_PICK(0, neg_x); RT_PICK(1, neg_y); RT_PICK(2, neg_z);
#undef RT_PICK
                const float nl[3] = {__uint_as_float(n_l[0]), __uint_as_float(n_l[1]), __uint_as_float(n_l[2])};
                const float fl[3] = {__uint_as_float(f_l[0]), __uint_as_float(f_l[1]), __uint_as_float(f_l[2])};
                const float nr[3] = {__uint_as_float(n_r[0]), __uint_as_float(n_r[1]), __uint_as_float(n_r[2])};
                const float fr[3] = {__uint_as_float(f_r[0]), __uint_as_float(f_r[1]), __uint_as_float(f_r[2])};
                rb.interval_picked(nl, fl, tl, tfl);
                rb.interval_picked(nr, fr, tr, tfr);
            } else {
                rb.interval(lo_l, hi_l, tl, tfl);
                rb.interval(lo_r, hi_r, tr, tfr);
            }
            // `mask` is the wave-uniform lane mask of the rays that reached this node, so "this lane's ray hits child k" as
            // a lane mask is one scalar AND with the compare's result: no per-lane flag, no flag-to-mask round trip
            const unsigned long long ml = mask & rb.accept_mask(tl, tfl);
            const unsigned long long mr = cr != REF_EMPTY ? mask & rb.accept_mask(tr, tfr) : 0ull;
            if (STATS && ((mask >> lane) & 1ull)) ++n_nodes;
            if (STATS) ++w_nodes;
            if (ml && mr) {
                // near child first, by majority of the lanes that care (a ray in both children votes for the nearer one)
                const unsigned long long both = ml & mr, le = __builtin_amdgcn_fcmpf(tl, tr, 5 /* LLVM FCMP_OLE */);
                const int vl = __popcll(ml & ~mr) + __popcll(both & le);
                const int vr = __popcll(mr & ~ml) + __popcll(both & ~le);
                const bool left_first = vl >= vr;
                const unsigned long long fm = left_first ? mr : ml;
                if (lane == 0) {
                    PacketEntry e;
                    e.ref = left_first ? cr : cl; e.mask_lo = (uint32_t)fm; e.mask_hi = (uint32_t)(fm >> 32);
                    stack[sp] = e;
                }
                ++sp;
                cur = left_first ? cl : cr;
                mask = left_first ? ml : mr;
                continue;
            } else if (ml) {
                cur = cl; mask = ml;
                continue;
            } else if (mr) {
                cur = cr; mask = mr;
                continue;
            }
            if (STATS) ++w_dead;
        } else {
            const uint32_t first = cur & REF_FIRST_MASK, count = ((cur >> 28) & 3u) + 1u;
            const bool mine = (mask >> lane) & 1ull;
            if (STATS && mine) n_prims += count;
            if (STATS) w_prims += count;
            if (cur & REF_SPHERE) {
                for (uint32_t k = 0; k < count; ++k) {
                    const u32x16 g = uload16(&S.sphs[first + k]);
                    const D3 c = mk(mkd(g[0], g[1]), mkd(g[2], g[3]), mkd(g[4], g[5]));
                    const double r = mkd(g[6], g[7]);
                    const D3 rot2 = mk(mkd(g[8], g[9]), mkd(g[10], g[11]), mkd(g[12], g[13]));
                    if (mine) sph_test_do(c, r, rot2, o, d, t_min, best, [&](double t) { best = t; prim = g[14]; rb.nearest(best); });
                }
            } else {
                for (uint32_t k = 0; k < count; ++k) {
                    const u32x16 g = uload16(&S.tris[first + k]);
                    const u32x4 g2 = uload4((const char*)&S.tris[first + k] + 64);
                    const D3 a = mk(mkd(g[0], g[1]), mkd(g[2], g[3]), mkd(g[4], g[5]));
                    const D3 e1 = mk(mkd(g[6], g[7]), mkd(g[8], g[9]), mkd(g[10], g[11]));
                    const D3 e2 = mk(mkd(g[12], g[13]), mkd(g[14], g[15]), mkd(g2[0], g2[1]));
                    if (mine) tri_test_do(a, e1, e2, o, d, t_min, best, [&](double t, double u, double v) { best = t; prim = g2[2]; bu = u; bv = v; rb.nearest(best); });
                }
            }
        }
        // pop (wave-uniform; LDS operations of one wave complete in order, so the read sees lane 0's write)
        if (sp == 0) break;
        --sp;
        const PacketEntry e = stack[sp];
        cur = __builtin_amdgcn_readfirstlane(e.ref);
        mask = ((unsigned long long)__builtin_amdgcn_readfirstlane(e.mask_hi) << 32) | __builtin_amdgcn_readfirstlane(e.mask_lo);
    }
    if (FUSE) {
        bool alive = false;
        D3 no = mk(0, 0, 0), nd = mk(0, 0, 0), nthr = mk(0, 0, 0);
        const uint32_t pid = blockIdx.x * TRACE_BLOCK + lane;
        if (active) {
            if (prim != RT_MISS) { hb.bu[i] = bu; hb.bv[i] = bv; }      // read back only by textured triangles
            alive = shade_path(S, pp, 0u, pid, ix.pixel, ix.local, i, o, d, mk(1.0, 1.0, 1.0), prim, best, hb, path_radiance, no, nd, nthr);
        }
        queue_append(alive, blockIdx.x, lane, qo, qg.shard_cap, out_counts, pid, no, nd, nthr);
    } else if (active) {
        hb.t[i] = prim == RT_MISS ? 0.0 : best;
        hb.prim[i] = prim;
        hb.bu[i] = bu;
        hb.bv[i] = bv;
    }
    if (STATS) {
        unsigned long long a = n_nodes, b = n_prims;
        for (int off = 32; off; off >>= 1) {
            a += __shfl_xor(a, off);
            b += __shfl_xor(b, off);
        }
        if (lane == 0) {
            atomicAdd(&counters[STAT_NODES], a);
            atomicAdd(&counters[STAT_PRIMS], b);
            atomicAdd(&counters[STAT_WAVE_NODES], (unsigned long long)w_nodes);
            atomicAdd(&counters[STAT_WAVE_PRIMS], (unsigned long long)w_prims);
            atomicAdd(&counters[STAT_LEAVES], (unsigned long long)w_dead);      // (this group's leaf-step slot is otherwise unused)

        }
    }
}

// ---------------------------------------------------------------------------------------------
// K2-K4 — shade + Russian roulette + compaction
// ---------------------------------------------------------------------------------------------
// One path vertex: sky on a miss, else the intersection record, Material::scatter and russian_roulette.
// Returns whether the path goes on (then no, nd, nthr are the next ray and throughput); a path that ends
// writes its radiance slot.  bu/bv are read lazily through hb (only textured triangles need them).
// pixel / local: the path's pixel index and local sample (decode_path of pid), for the RNG key.
__device__ __forceinline__ bool shade_path(const DScene& S, const PassParams& pp, uint32_t depth, uint32_t pid, uint32_t pixel,
                                           uint32_t local, uint32_t i, const D3 o, const D3 d, const D3 thr, uint32_t prim, double t,
                                           const HitBuf& hb, double* path_radiance, D3& no, D3& nd, D3& nthr) {
    bool alive = false;
    D3 L = mk(0.0, 0.0, 0.0);
    if (prim == RT_MISS) {
        // Camera::raycast miss branch (src/camera.rs:308-312)
        D3 dir = normalize(d);
        L = mul(thr, sky_color_toward(S.sky, dir));
    } else {
        SurfaceHit h;
        h.bu = &hb.bu[i]; h.bv = &hb.bv[i];
        h.is_sphere = prim < S.n_spheres;
        h.ts = nullptr; h.ss = nullptr;
        if (h.is_sphere) {                                      // Sphere::hit tail (src/hittable.rs:340-346)
            const SphShade* ss = &S.sph_shade[prim];
            h.ss = ss;
            h.p = add(scale(d, t), o);
            h.n = divs(sub(h.p, ld3(ss->c)), ss->r);
            h.front = dot(d, h.n) < 0.0;
            if (!h.front) h.n = neg(h.n);
            h.material = ss->material;
        } else {                                                // Triangle::hit tail (:463-464)
            const TriShade* ts = &S.tri_shade[prim - S.n_spheres];
            h.ts = ts;
            h.p = add(o, scale(d, t));
            h.n = ld3(ts->n);
            h.front = dot(d, h.n) <= 0.0;
            h.material = ts->material;
        }
        const ShadeMat* m = &S.mats[h.material];
        const uint32_t kind = m->kind, has_fuzz = m->has_fuzz;
        RngKey key = {pp.seed, pixel, pp.sample_offset + (pp.pass_base + local) * pp.sample_stride, depth};
        // Slots of a vertex: 0 roulette, 1-3 random_unit, 4 dielectric noise (Philox block b holds slots 2b, 2b+1): a
        // Lambertian or fuzzy-metal vertex needs blocks 0 and 1, a plain metal block 0, a clear dielectric block 2 only.
        D3 att, dir;
        U4 b0 = {0u, 0u, 0u, 0u};
        if (kind == RT_MAT_DIELECTRIC) {                        // src/material.rs:150-179
            double ior = m->ior;
            double ri = h.front ? 1.0 / ior : ior;
            D3 din = normalize(d);
            double c = fmin(-dot(din, h.n), 1.0);
            double sn = sqrt(1.0 - c * c);
            bool cannot = ri * sn > 1.0;
            U4 b2 = rng_block(key, 2);
            double noise = u64_to_unit(b2.x, b2.y);             // slot 4
            if (cannot || reflectance(c, ri) > noise) {
                dir = reflect(din, h.n);
            } else {
                dir = refract(din, h.n, ri);
                if (has_fuzz) {
                    b0 = rng_block(key, 0);
                    U4 b1 = rng_block(key, 1);
                    D3 ru = normalize(mk(2.0 * u64_to_unit(b0.z, b0.w) - 1.0, 2.0 * u64_to_unit(b1.x, b1.y) - 1.0,
                                         2.0 * u64_to_unit(b1.z, b1.w) - 1.0));
                    dir = add(dir, scale(ru, m->fuzz));
                }
            }
            dir = normalize(dir);
            att = mk(1.0, 1.0, 1.0);
        } else {
            D3 ru = mk(0, 0, 0);
            b0 = rng_block(key, 0);
            if (kind == RT_MAT_LAMBERTIAN || has_fuzz) {        // Vec3::random_unit, slots 1,2,3 (src/vec3.rs:113-120)
                U4 b1 = rng_block(key, 1);
                ru = normalize(mk(2.0 * u64_to_unit(b0.z, b0.w) - 1.0, 2.0 * u64_to_unit(b1.x, b1.y) - 1.0,
                                  2.0 * u64_to_unit(b1.z, b1.w) - 1.0));
            }
            if (kind == RT_MAT_LAMBERTIAN) {                    // src/material.rs:110-120
                dir = add(h.n, ru);
                if (near_zero(dir)) dir = h.n;
            } else {                                            // Metal (:94-108)
                dir = reflect(d, h.n);
                if (has_fuzz) dir = add(dir, scale(ru, m->fuzz));
            }
            att = m->tex_kind == RT_TEX_SOLID ? ld3(m->rgb) : texture_value(S, m->texture, h);
        }
        dir = normalize(dir);                                   // Ray::new(point, dir)
        if (depth < pp.cam.max_depth) {                         // src/camera.rs:300
            double p = fmax(fmax(att.x, att.y), att.z);         // russian_roulette (:280-293)
            // slot 0; u < 1 whatever is drawn, so p >= 1 (every dielectric: att = ONE) needs no draw — and b0 is only
            // missing for dielectrics that did not need random_unit
            if (p >= 1.0 || u64_to_unit(b0.x, b0.y) < p) {
                alive = true;
                nthr = mul(thr, divs(scale(att, 1.0), p));
                no = h.p;
                nd = dir;
            }
        }
    }
    if (!alive) {
        path_radiance[3 * (size_t)pid + 0] = L.x;
        path_radiance[3 * (size_t)pid + 1] = L.y;
        path_radiance[3 * (size_t)pid + 2] = L.z;
    }
    return alive;
}

// K4: survivors of logical wave `wave` are appended to its output shard: wave64 ballot + prefix count, one
// atomic per wave on that shard's counter (counters sit 128 B apart).  Called by all 64 lanes.
__device__ __forceinline__ void queue_append(bool alive, uint32_t wave, uint32_t lane, const RayQueue& qo, uint32_t shard_cap,
                                             uint32_t* out_counts, uint32_t pid, const D3 no, const D3 nd, const D3 nthr) {
    const unsigned long long mask = __ballot(alive);
    if (mask) {
        const uint32_t shard = wave & (NSHARD - 1);
        uint32_t base = 0;
        if (lane == 0) base = atomicAdd(out_counts + COUNT_STRIDE * shard, (uint32_t)__popcll(mask));
        base = __shfl(base, 0) + shard * shard_cap;
        if (alive) {
            const uint32_t j = base + (uint32_t)__popcll(mask & ((1ull << lane) - 1ull));
            qo.o[0][j] = no.x; qo.o[1][j] = no.y; qo.o[2][j] = no.z;
            qo.d[0][j] = nd.x; qo.d[1][j] = nd.y; qo.d[2][j] = nd.z;
            qo.thr[0][j] = nthr.x; qo.thr[1][j] = nthr.y; qo.thr[2][j] = nthr.z;
            qo.pid[j] = pid;
        }
    }
}

// One launched wave per logical wave of the queue.  The host does not know how many entries the queue holds (the
// counts stay on the device): it launches k_shade for its ESTIMATE of the queue's logical waves (the sizes the previous
// passes reported through `hist`) and k_shade_rest, a small persistent grid, for every wave beyond that — whatever the
// estimate, every entry is shaded exactly once.  Without an estimate k_shade covers the upper bound and the waves beyond
// the queue's end leave at once.  hist (may be null): wave 0 stores the queue's entry count for the host's next estimate.
// (The persistent grid-stride form needs 147 instead of 90 VGPRs — three instead of five waves per SIMD — so it only
// takes the overflow.)
template <bool LIGHT>
__device__ __forceinline__ void shade_wave(const DScene& S, const PassParams& pp, uint32_t depth, const RayQueue& qi, const QueueGeom& qg,
                                           const WaveQueue& wq, const HitBuf& hb, const RayQueue& qo, uint32_t* out_counts,
                                           double* path_radiance, uint32_t wave, uint32_t lane) {
    uint32_t i = 0;
    const bool active = LIGHT ? wq.slot_light(qg, wave, lane, i) : wq.slot(wave, lane, i);
    bool alive = false;
    D3 no = mk(0, 0, 0), nd = mk(0, 0, 0), nthr = mk(0, 0, 0);
    uint32_t pid = 0;
    if (active) {
        pid = qi.pid[i];
        const D3 o = mk(qi.o[0][i], qi.o[1][i], qi.o[2][i]);
        const D3 d = mk(qi.d[0][i], qi.d[1][i], qi.d[2][i]);
        // depth 0: the throughput is (1,1,1) by construction (k_raygen writes it, nobody needs to read it)
        const D3 thr = depth == 0 ? mk(1.0, 1.0, 1.0) : mk(qi.thr[0][i], qi.thr[1][i], qi.thr[2][i]);
        const uint32_t prim = hb.prim[i];
        const double t = prim == RT_MISS ? 0.0 : hb.t[i];
        const PathIndex ix = decode_path(pp, pid);
        alive = shade_path(S, pp, depth, pid, ix.pixel, ix.local, i, o, d, thr, prim, t, hb, path_radiance, no, nd, nthr);
    }
    queue_append(alive, wave, lane, qo, wq.shard_cap, out_counts, pid, no, nd, nthr);
}
#ifdef RT_SHADE_WAVES
#define RT_SHADE_ATTR __attribute__((amdgpu_waves_per_eu(RT_SHADE_WAVES, RT_SHADE_WAVES)))
#else
#define RT_SHADE_ATTR
#endif
__global__ RT_SHADE_ATTR __launch_bounds__(BLOCK) void k_shade(DScene S, PassParams pp, uint32_t depth, RayQueue qi, QueueGeom qg,
                                                 HitBuf hb, RayQueue qo, uint32_t* out_counts, double* path_radiance, uint32_t* hist) {
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t wave = __builtin_amdgcn_readfirstlane((blockIdx.x * BLOCK + threadIdx.x) >> 6);
    // A wave of this kernel lives for a few microseconds, so its start-up chain matters: the two words of the queue's
    // counters it needs (size in waves — left by the traversal kernel —, its own shard's count) and the ray + hit record of
    // its slot are all requested at once; the slot's address does not depend on the counts, and every slot of a shard is
    // allocated memory, so the loads are safe even where the entry turns out not to exist.
    const uint32_t shard = wave & (NSHARD - 1);
    const uint32_t off = (wave / NSHARD) * 64u + lane;
    const uint32_t i = shard * qg.shard_cap + off;
    const bool direct = qg.count == nullptr;
    const uint32_t pid = qi.pid[i];
    const D3 o = mk(qi.o[0][i], qi.o[1][i], qi.o[2][i]);
    const D3 d = mk(qi.d[0][i], qi.d[1][i], qi.d[2][i]);
    // depth 0: the throughput is (1,1,1) by construction (k_raygen writes it, nobody needs to read it)
    const D3 thr = depth == 0 ? mk(1.0, 1.0, 1.0) : mk(qi.thr[0][i], qi.thr[1][i], qi.thr[2][i]);
    const uint32_t prim = hb.prim[i];
    const double t_hit = hb.t[i];
    // (after the vector loads in program order: scalar loads return out of order, so waiting for these two means waiting
    // for all of them — issued first they would hold the ray loads back by a round trip)
    uint32_t n_waves = (qg.n_direct + 63u) >> 6, c_shard = 0;
    if (!direct) { n_waves = qg.count[QUEUE_NWAVES_WORD]; c_shard = qg.count[COUNT_STRIDE * shard]; }
    if (hist && wave == 0) {           // the queue's entry count, for the host's next launch estimate
        uint32_t tot = direct ? 0u : qg.count[COUNT_STRIDE * lane];
        for (int sh = 32; sh; sh >>= 1) tot += (uint32_t)__shfl_xor((int)tot, sh);
        if (lane == 0) *hist = direct ? qg.n_direct : tot;
    }
    if (wave >= n_waves) return;
    const bool active = direct ? wave * 64u + lane < qg.n_direct : off < c_shard;
    bool alive = false;
    D3 no = mk(0, 0, 0), nd = mk(0, 0, 0), nthr = mk(0, 0, 0);
    if (active) {
        const PathIndex ix = decode_path(pp, pid);
        alive = shade_path(S, pp, depth, pid, ix.pixel, ix.local, i, o, d, thr, prim, prim == RT_MISS ? 0.0 : t_hit, hb, path_radiance, no, nd, nthr);
    }
    queue_append(alive, wave, lane, qo, qg.shard_cap, out_counts, active ? pid : 0u, no, nd, nthr);
}
__global__ __launch_bounds__(BLOCK) void k_shade_rest(DScene S, PassParams pp, uint32_t depth, RayQueue qi, QueueGeom qg,
                                                      HitBuf hb, RayQueue qo, uint32_t* out_counts, double* path_radiance, uint32_t first_wave) {
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t wave0 = __builtin_amdgcn_readfirstlane((blockIdx.x * BLOCK + threadIdx.x) >> 6);
    const uint32_t n_launch = gridDim.x * (BLOCK / 64);
    WaveQueue wq;
    wq.load(qg, lane);
    for (uint32_t wave = first_wave + wave0; wave < wq.n_waves; wave += n_launch)
        shade_wave<false>(S, pp, depth, qi, qg, wq, hb, qo, out_counts, path_radiance, wave, lane);
}

// ---------------------------------------------------------------------------------------------
// K5 — per-pixel sum in sample order, and the final mean
// ---------------------------------------------------------------------------------------------
// zero_counts (may be null): the pass's queue counters, cleared here for the next pass on this workspace — every kernel
// that reads or appends to them has finished by the time the pass's sums are taken.
__global__ __launch_bounds__(BLOCK) void k_accumulate(const double* path_radiance, uint32_t n_pixels, uint32_t W, uint32_t H,
                                                      uint32_t tiled, uint32_t group, uint32_t n_local, double* rgb_sum,
                                                      uint32_t* zero_counts, uint32_t n_zero) {
    uint32_t slot = blockIdx.x * BLOCK + threadIdx.x;
    if (zero_counts && slot < n_zero) zero_counts[slot] = 0u;
    if (slot >= n_pixels) return;
    const uint32_t pixel = slot_to_pixel(slot, W, H, tiled);
    double r = rgb_sum[3 * (size_t)pixel], g = rgb_sum[3 * (size_t)pixel + 1], b = rgb_sum[3 * (size_t)pixel + 2];
    for (uint32_t s = 0; s < n_local; ++s) {
        const double* L = &path_radiance[3 * slot_sample_to_pid(slot, s, n_pixels, group)];
        r += L[0]; g += L[1]; b += L[2];
    }
    rgb_sum[3 * (size_t)pixel] = r; rgb_sum[3 * (size_t)pixel + 1] = g; rgb_sum[3 * (size_t)pixel + 2] = b;
}

// The same sums for the usual layout (16 samples of a pixel per wave of camera rays, rt_device.h: slot_sample_to_pid):
// there the 64 paths with consecutive ids are 16 samples of 4 neighbouring pixel slots, so a thread of k_accumulate that
// follows ONE pixel through its samples reads 24 bytes here, 24 bytes 96 bytes further on, ...  Here a wave takes such a
// block of 64 paths as it lies — 1536 contiguous bytes, three fully coalesced loads — through LDS, and twelve lanes
// (4 pixels x 3 channels) add their 16 values in ascending sample order, sample group after sample group: the same
// additions in the same order as k_accumulate, so the frame is the same bit for bit.  One small workgroup per block of
// four pixels: 364 -> 327 us per pass at 1080p.  (A grid of resident waves striding over the blocks with the next block's
// loads in flight streams faster on its own but cost the frame 6 %: the passes' sums are ordered through events, and
// long-lived waves of one pass's sum kernel stand in the way of the next pass's.)
__global__ __launch_bounds__(64) void k_accumulate16(const double* path_radiance, uint32_t n_pixels, uint32_t W, uint32_t H, uint32_t n_local,
                                                     double* rgb_sum, uint32_t* zero_counts, uint32_t n_zero) {
    __shared__ double blk[192];
    const uint32_t lane = threadIdx.x, b = blockIdx.x;                  // b: block of four pixel slots 4b .. 4b+3
    const uint32_t gid = b * 64u + lane;
    if (zero_counts && gid < n_zero) zero_counts[gid] = 0u;
    if (4u * b >= n_pixels) return;                                     // (wave-uniform)
    const uint32_t j = lane / 3u, c = lane - 3u * j;                    // lanes 0..11: pixel slot 4b + j, channel c
    size_t at = 0;
    double acc = 0.0;
    if (lane < 12u) {
        at = 3 * (size_t)slot_to_pixel(4u * b + j, W, H, 1u) + c;
        acc = rgb_sum[at];
    }
    const size_t group_stride = (size_t)n_pixels * 16u * 3u;            // doubles between two sample groups of the same block
    const double* src = path_radiance + (size_t)b * 192u;
    for (uint32_t sg = 0; sg < n_local / 16u; ++sg, src += group_stride) {
        const double v0 = src[lane], v1 = src[lane + 64u], v2 = src[lane + 128u];
        __syncthreads();                                                // (one wave: the barrier only orders the LDS traffic)
        blk[lane] = v0; blk[lane + 64u] = v1; blk[lane + 128u] = v2;
        __syncthreads();
        if (lane < 12u) {
#pragma unroll
            for (uint32_t s_ = 0; s_ < 16u; ++s_) acc += blk[(4u * s_ + j) * 3u + c];      // path 4 s + j of the block: sample s of pixel j
        }
    }
    if (lane < 12u) rgb_sum[at] = acc;
}

__global__ __launch_bounds__(BLOCK) void k_resolve(const double* sum, double* mean, size_t n, double total_spp) {
    size_t i = (size_t)blockIdx.x * BLOCK + threadIdx.x;
    if (i < n) mean[i] = sum[i] / total_spp;             // `/ num_samples as Float` (src/camera.rs:324)
}

// The f32 form of the multi-GPU exchange ("ncclReduce(sum, f32, W*H*3)"): a device's f64 sums are narrowed
// once, the f32 buffers are summed over xGMI (half the bytes), the root divides.  Only the cross-device sum is f32.
__global__ __launch_bounds__(BLOCK) void k_narrow_f32(const double* in, float* out, size_t n) {
    const size_t i = (size_t)blockIdx.x * BLOCK + threadIdx.x;
    if (i < n) out[i] = (float)in[i];
}
__global__ __launch_bounds__(BLOCK) void k_resolve_f32(const float* sum, double* mean, size_t n, double total_spp) {
    const size_t i = (size_t)blockIdx.x * BLOCK + threadIdx.x;
    if (i < n) mean[i] = (double)sum[i] / total_spp;
}

// f64::round (half away from zero, like C's round) of a value already in [0, 255]
__device__ __forceinline__ uint8_t round_u8(double x) { return (uint8_t)round(x); }
__device__ __forceinline__ double clamp_colour(double c) {      // the reference panics outside [0, 1] (src/vec3.rs:44-63): clamp; NaN -> 0
    if (!(c >= 0.0)) c = 0.0;
    return c > 1.0 ? 1.0 : c;
}

// Output stage on the device: Vec3Ext::as_rgb_gamma (src/vec3.rs:71-97) over a resolved frame,
// round(pow(c, 1/2.2) * 255) per channel, clamped like the host routine rt_encode_gamma8.
__global__ __launch_bounds__(BLOCK) void k_gamma8(const double* mean, size_t n, uint8_t* out) {
    const size_t i = (size_t)blockIdx.x * BLOCK + threadIdx.x;
    if (i >= n) return;
    const double gamma = 1.0 / 2.2;                       // as_gamma_vec (:39-42)
    out[i] = round_u8(pow(clamp_colour(mean[i]), gamma) * 255.0);
}

// One pass of the preview loop's blend (window::render_thread, src/window.rs:277-310) for every pixel:
//   new = pass_sum / n  (render_pixel's mean);  combined = new * (n / total) + old * (1 - n / total)
// quantise: old is the 8-bit LINEAR frame / 255 and the result goes back through as_rgb_linear (round(c * 255)) — the
// reference's accumulation; otherwise old is the float frame kept beside it (no re-quantisation error).  A = 0xff.
__global__ __launch_bounds__(BLOCK) void k_blend_pass(const double* pass_sum, double* mean_frame, uint8_t* rgba8, size_t n_pixels,
                                                      double n_samples, double total_samples, uint32_t quantise) {
    const size_t px = (size_t)blockIdx.x * BLOCK + threadIdx.x;
    if (px >= n_pixels) return;
    const double new_ratio = n_samples / total_samples;   // :293
    const double old_ratio = 1.0 - new_ratio;             // :294
    uint8_t o[4];
    for (int c = 0; c < 3; ++c) {
        const double nw = pass_sum[3 * px + c] / n_samples;                           // src/camera.rs:324
        const double old = quantise ? (double)rgba8[4 * px + c] / 255.0 : mean_frame[3 * px + c];
        const double comb = (nw * new_ratio) + (old * old_ratio);                     // :295
        mean_frame[3 * px + c] = comb;
        o[c] = round_u8(clamp_colour(comb) * 255.0);                                 // as_rgb_linear (src/vec3.rs:64-68)
    }
    o[3] = 0xff;
    *reinterpret_cast<uchar4*>(rgba8 + 4 * px) = make_uchar4(o[0], o[1], o[2], o[3]);
}

__global__ __launch_bounds__(BLOCK) void k_pack_hits(HitBuf hb, QueueGeom qg, rt_hit* out) {
    const uint32_t p = blockIdx.x * BLOCK + threadIdx.x;
    uint32_t i = 0;
    if (!direct_slot(qg, p, i)) return;
    rt_hit h;
    h.t = hb.t[i]; h.prim = hb.prim[i]; h._pad = 0;
    out[p] = h;
}
__global__ __launch_bounds__(BLOCK) void k_pack_rays(RayQueue q, QueueGeom qg, rt_ray* out) {
    const uint32_t p = blockIdx.x * BLOCK + threadIdx.x;
    uint32_t i = 0;
    if (!direct_slot(qg, p, i)) return;
    rt_ray r;
    for (int k = 0; k < 3; ++k) { r.o[k] = q.o[k][i]; r.d[k] = q.d[k][i]; }
    out[p] = r;
}

// ---------------------------------------------------------------------------------------------
// launchers
// ---------------------------------------------------------------------------------------------
FastDiv fast_div_for(uint32_t d) {
    FastDiv f;
    if (d == 0) d = 1;
    uint32_t L = 0;
    while ((1ull << L) < d) ++L;                           // ceil(log2 d)
    f.m = (uint32_t)(((1ull << 32) * ((1ull << L) - d)) / d + 1);
    f.s1 = L < 1 ? L : 1;
    f.s2 = L > 0 ? L - 1 : 0;
    f.d = d;
    return f;
}
void finish_pass_params(PassParams& pp) {
    const uint32_t G = pp.group <= 1u ? 1u : pp.group;
    uint32_t lp = 0;
    while ((64u >> lp) > G) ++lp;                          // 64 / G = 2^lp
    pp.log2_p = G == 1u ? 6u : lp;
    pp.div_group = fast_div_for(pp.group <= 1u ? std::max(pp.n_pixels, 1u) : pp.n_pixels * G);
    pp.div_tpr = fast_div_for(std::max((pp.cam.width & ~7u) >> 3, 1u));
    pp.div_width = fast_div_for(std::max(pp.cam.width, 1u));
}

static inline unsigned grid_for(size_t n) { return (unsigned)((n + BLOCK - 1) / BLOCK); }

void launch_jitter_table(const PassParams& pp, uint32_t n_local, double* d_out, hipStream_t s) {
    if (!n_local) return;
    hipLaunchKernelGGL(k_jitter_table, dim3(grid_for(n_local)), dim3(BLOCK), 0, s, pp.sample_offset, pp.sample_stride, pp.pass_base, n_local, d_out);
}
void launch_raygen(const PassParams& pp, const QueueGeom& qg, RayQueue q, hipStream_t s) {
    if (!qg.n_direct) return;
    hipLaunchKernelGGL(k_raygen, dim3(grid_for(qg.n_direct)), dim3(BLOCK), 0, s, pp, qg, q);
}
void launch_prepare_rays(const rt_ray* rays, const QueueGeom& qg, RayQueue q, hipStream_t s) {
    if (!qg.n_direct) return;
    hipLaunchKernelGGL(k_prepare_rays, dim3(grid_for(qg.n_direct)), dim3(BLOCK), 0, s, rays, qg, q);
}
// waves the chip holds at once of a kernel (one wave per workgroup) with `lds` bytes of dynamic LDS
// (max_per_cu: a cap on the waves per CU to count, 0 = none)
template <class K>
static unsigned resident_waves(K kernel, size_t lds, unsigned max_per_cu = 0) {
    int dev = 0, per_cu = 0;
    hipDeviceProp_t pr;
    if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&pr, dev) != hipSuccess ||
        hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, TRACE_BLOCK, lds) != hipSuccess || per_cu < 1)
        return 256u * 8u;
    if (max_per_cu && (unsigned)per_cu > max_per_cu) per_cu = (int)max_per_cu;
    return (unsigned)pr.multiProcessorCount * (unsigned)per_cu;
}
// n_upper: an upper bound of the queue's entries the host knows (the pass size): the kernel reads the real counts itself
template <class BoxT, int WIDTH, bool STATS, bool SHADE, int MODE>
static void launch_walk_k(LaunchCtx& ctx, const DScene& S, RayQueue q, const QueueGeom& qg, uint32_t n_upper, uint32_t n_estimate, double t_min,
                          double t_max, HitBuf hb, unsigned long long* d_counters, const WalkShade& sh, uint32_t stack_entries,
                          uint32_t ovf_waves, hipStream_t s) {
    const size_t lds = (size_t)stack_entries * TRACE_BLOCK * sizeof(uint32_t);
    // the occupancy query is per (kernel instance, stack size); a scene runs one box format, width and mode, and a handful
    // of stack sizes occur: the answers are kept with the scene (one device), filled under the device's mutex
    const int ci = stack_entries <= 16 ? 0 : stack_entries <= 24 ? 1 : stack_entries <= 32 ? 2 : stack_entries <= 48 ? 3 : 4;
    constexpr bool OVF = MODE == WALK_OVF;
#ifdef RT_NO_FIVE      // (A/B builds)
    constexpr bool FIVE_OK = false;
#else
    constexpr bool FIVE_OK = !SHADE && !STATS && WIDTH == 4 && std::is_same<BoxT, Box16>::value;      // (k_walk5)
#endif
    auto kernel = [] { if constexpr (SHADE) return k_walk_shade<BoxT, WIDTH, STATS, MODE>; else return k_walk<BoxT, WIDTH, STATS, MODE>; }();
    // (the answers for "several passes in flight" are kept apart: another kernel, or the same one without the cap below)
    const bool shared = ctx.five && !SHADE && !STATS;
    unsigned* slot_cache = shared ? &ctx.walk5_slots[ci] : &ctx.walk_slots[STATS ? 1 : 0][SHADE ? 1 : 0][ci];
    if constexpr (FIVE_OK) {
        if (ctx.five) kernel = k_walk5<BoxT, WIDTH, STATS, MODE>;
    }
    unsigned& cached = *slot_cache;
    if (!cached) {
        // A stand-alone walk that has the GPU to itself (one pass in flight, rt_trace_closest) runs best with FOUR waves per
        // SIMD although its 92 registers admit five: 85.5 against 90.7 ms per frame of the 1 M-triangle scene with one
        // pass in flight.  With several passes in flight the SIMDs are shared anyway and the full count is asked for.
        const bool alone = !SHADE && !STATS && !ctx.five;
        cached = resident_waves(kernel, lds, alone ? 16u : 0u);
        if (ctx.debug) fprintf(stderr, "[rt_hip] walk kernel (stats %d, shade %d, mode %d, 96 registers %d): %zu B of LDS per wave, %u resident waves\n", (int)STATS, (int)SHADE, MODE, (int)(FIVE_OK && ctx.five), lds, cached);
    }
    const unsigned chunks = (n_upper + 63u) / 64u;
    // persistent (one wave per hardware slot, lanes refilled); never more waves than the queue can have chunks.
    unsigned hw = cached;
    // With an estimate of the queue's size (what the previous passes reported) the grid is sized for it — a wave per 64
    // expected entries, a quarter more for safety; the waves are persistent, so any real size is still served in full.
    if (n_estimate) hw = std::min<unsigned>(hw, std::max<unsigned>(64u, (n_estimate + n_estimate / 4 + 63u) / 64u));
    // (tests: one-chunk-per-wave launches instead of persistent ones, or a given number of persistent waves)
    unsigned slots = ctx.refill ? std::min(ctx.refill_slots ? ctx.refill_slots : hw, chunks) : chunks;
    if (OVF) slots = std::min<unsigned>(slots, ovf_waves);      // the overflow rows have one column per launched lane
    if (!slots) return;
    hipLaunchKernelGGL(kernel, dim3(slots), dim3(TRACE_BLOCK), lds, s, S, q, qg, t_min, t_max, hb, d_counters, sh);
}
template <class BoxT, int WIDTH>
static void launch_walk_b(LaunchCtx& ctx, const DScene& S, RayQueue q, const QueueGeom& qg, uint32_t n_upper, uint32_t n_estimate, double t_min, double t_max,
                          HitBuf hb, bool stats, unsigned long long* d_counters, const WalkShade* sh, const StackOverflow& ovf,
                          hipStream_t s) {
    // The per-lane stack needs S.stack4 entries for the BVH4 and one entry per level for the BVH2; a smaller stack means
    // less LDS per wave, i.e. more waves per CU.
    const uint32_t need = WIDTH == 4 ? S.stack4 : S.bvh_depth;
    const uint32_t entries = need <= 16 ? 16 : need <= 24 ? 24 : need <= 32 ? 32 : need <= 48 ? 48 : 64;
    WalkShade w;
    if (sh) w = *sh; else memset(&w, 0, sizeof(w));
    w.stack_ovf = nullptr;
    auto go = [&](auto mode, uint32_t rows, uint32_t ovf_waves) {
        constexpr int M = decltype(mode)::value;
        if (sh) {
            if (stats) launch_walk_k<BoxT, WIDTH, true, true, M>(ctx, S, q, qg, n_upper, n_estimate, t_min, t_max, hb, d_counters, w, rows, ovf_waves, s);
            else launch_walk_k<BoxT, WIDTH, false, true, M>(ctx, S, q, qg, n_upper, n_estimate, t_min, t_max, hb, d_counters, w, rows, ovf_waves, s);
        } else {
            if (stats) launch_walk_k<BoxT, WIDTH, true, false, M>(ctx, S, q, qg, n_upper, n_estimate, t_min, t_max, hb, d_counters, w, rows, ovf_waves, s);
            else launch_walk_k<BoxT, WIDTH, false, false, M>(ctx, S, q, qg, n_upper, n_estimate, t_min, t_max, hb, d_counters, w, rows, ovf_waves, s);
        }
    };
    if constexpr (WIDTH == 4) {
        // deep trees: 32 rows in LDS, the rest in the caller's overflow buffer (persistent launches only: a column per launched lane)
        const uint32_t lds_rows = ctx.ovf_lds_rows;
        if (need > lds_rows && ovf.rows && need <= lds_rows + ovf.rows_cap && ovf.waves && ctx.refill) {
            w.stack_ovf = ovf.rows;
            w.ovf_lds_rows = lds_rows;
            go(std::integral_constant<int, WALK_OVF>(), lds_rows, ovf.waves);
            return;
        }
        if (S.all_lanes_load) {
            go(std::integral_constant<int, WALK_ALL>(), entries, 0u);
            return;
        }
    }
    go(std::integral_constant<int, WALK_PLAIN>(), entries, 0u);
}
// the BVH8 walk: the whole stack in LDS (two words per level of internal nodes)
template <bool STATS, bool SHADE>
static void launch_walk8_k(LaunchCtx& ctx, const DScene& S, RayQueue q, const QueueGeom& qg, uint32_t n_upper, uint32_t n_estimate, double t_min,
                           double t_max, HitBuf hb, unsigned long long* d_counters, const WalkShade& sh, hipStream_t s) {
    const uint32_t need = std::max(S.depth8, 1u);
    const uint32_t entries = need <= 8 ? 8 : need <= 12 ? 12 : need <= 16 ? 16 : (uint32_t)BVH8_MAX_DEPTH;
    const int ci = entries <= 8 ? 0 : entries <= 12 ? 1 : entries <= 16 ? 2 : 3;
    const size_t lds = (size_t)2 * entries * TRACE_BLOCK * sizeof(uint32_t);
    auto kernel = [] { if constexpr (SHADE) return k_walk8_shade<STATS>; else return k_walk8<STATS>; }();
    unsigned& cached = ctx.walk_slots[STATS ? 1 : 0][SHADE ? 1 : 0][ci];
    if (!cached) cached = resident_waves(kernel, lds);
    const unsigned chunks = (n_upper + 63u) / 64u;
    unsigned hw = cached;
    if (n_estimate) hw = std::min<unsigned>(hw, std::max<unsigned>(64u, (n_estimate + n_estimate / 4 + 63u) / 64u));
    const unsigned slots = ctx.refill ? std::min(ctx.refill_slots ? ctx.refill_slots : hw, chunks) : chunks;
    if (!slots) return;
    hipLaunchKernelGGL(kernel, dim3(slots), dim3(TRACE_BLOCK), lds, s, S, q, qg, t_min, t_max, hb, d_counters, sh);
}
static void launch_walk_any(LaunchCtx& ctx, const DScene& S, RayQueue q, const QueueGeom& qg, uint32_t n_upper, uint32_t n_estimate, double t_min, double t_max,
                            HitBuf hb, bool stats, unsigned long long* d_counters, const WalkShade* sh, const StackOverflow& ovf,
                            hipStream_t s) {
    if (!n_upper) return;
    if (S.pool8 && S.depth8 <= (uint32_t)BVH8_MAX_DEPTH) {
        WalkShade w;
        if (sh) w = *sh; else memset(&w, 0, sizeof(w));
        w.stack_ovf = nullptr;
        if (sh) {
            if (stats) launch_walk8_k<true, true>(ctx, S, q, qg, n_upper, n_estimate, t_min, t_max, hb, d_counters, w, s);
            else launch_walk8_k<false, true>(ctx, S, q, qg, n_upper, n_estimate, t_min, t_max, hb, d_counters, w, s);
        } else {
            if (stats) launch_walk8_k<true, false>(ctx, S, q, qg, n_upper, n_estimate, t_min, t_max, hb, d_counters, w, s);
            else launch_walk8_k<false, false>(ctx, S, q, qg, n_upper, n_estimate, t_min, t_max, hb, d_counters, w, s);
        }
        return;
    }
    if (S.stack4 <= (uint32_t)BVH4_MAX_STACK) {
        if (S.nodes4) { launch_walk_b<Box16, 4>(ctx, S, q, qg, n_upper, n_estimate, t_min, t_max, hb, stats, d_counters, sh, ovf, s); return; }
        if (S.nodes4f) { launch_walk_b<Box32, 4>(ctx, S, q, qg, n_upper, n_estimate, t_min, t_max, hb, stats, d_counters, sh, ovf, s); return; }
    }
    if (S.nodes16)
        launch_walk_b<Box16, 2>(ctx, S, q, qg, n_upper, n_estimate, t_min, t_max, hb, stats, d_counters, sh, ovf, s);
    else if (S.nodes32)
        launch_walk_b<Box32, 2>(ctx, S, q, qg, n_upper, n_estimate, t_min, t_max, hb, stats, d_counters, sh, ovf, s);
    else
        launch_walk_b<Box64, 2>(ctx, S, q, qg, n_upper, n_estimate, t_min, t_max, hb, stats, d_counters, sh, ovf, s);
}
uint32_t stack_overflow_rows(const DScene& S, uint32_t lds_rows) {
    const bool w4 = !S.pool8 && S.stack4 <= (uint32_t)BVH4_MAX_STACK && (S.nodes4 || S.nodes4f);
    return (w4 && S.stack4 > lds_rows) ? S.stack4 - lds_rows : 0u;
}
void launch_walk(LaunchCtx& ctx, const DScene& S, RayQueue q, const QueueGeom& qg, uint32_t n_upper, double t_min, double t_max, HitBuf hb,
                 bool stats, unsigned long long* d_counters, hipStream_t s, uint32_t n_estimate, const StackOverflow* ovf) {
    StackOverflow none = {nullptr, 0u, 0u};
    launch_walk_any(ctx, S, q, qg, n_upper, n_estimate, t_min, t_max, hb, stats, d_counters, nullptr, ovf ? *ovf : none, s);
}
void launch_walk_shade(LaunchCtx& ctx, const DScene& S, const PassParams& pp, uint32_t depth0, RayQueue q, const QueueGeom& qg,
                       uint32_t n_upper, double t_min, double t_max, HitBuf hb, double* d_path_radiance, uint32_t* d_hist, bool stats,
                       unsigned long long* d_counters, hipStream_t s, uint32_t n_estimate, const StackOverflow* ovf) {
    WalkShade sh;
    memset(&sh, 0, sizeof(sh));
    sh.pp = pp;
    sh.depth0 = depth0;
    sh.path_radiance = d_path_radiance;
    sh.hist = d_hist;
    StackOverflow none = {nullptr, 0u, 0u};
    launch_walk_any(ctx, S, q, qg, n_upper, n_estimate, t_min, t_max, hb, stats, d_counters, &sh, ovf ? *ovf : none, s);
}

bool packet_supported(const DScene& S) { return S.nodes32 != nullptr; }   // f64-box scenes have no packet variant

void launch_trace_packet(const DScene& S, const PassParams* gen, RayQueue q, const QueueGeom& qg, double t_min, double t_max,
                         HitBuf hb, bool stats, unsigned long long* d_counters, hipStream_t s, const PacketFuse* fuse) {
    if (!qg.n_direct) return;
    const dim3 grid((qg.n_direct + TRACE_BLOCK - 1) / TRACE_BLOCK), block(TRACE_BLOCK);
    PassParams pp;
    if (gen) pp = *gen; else memset(&pp, 0, sizeof(pp));
    PacketFuse f;
    if (fuse && gen) f = *fuse; else memset(&f, 0, sizeof(f));
    // (a BVH4 variant of this kernel — four boxes per s_load, stack in VGPR lanes — was measured 8 % slower:
    // the wave tests all four boxes of every node on all lanes, and the node steps were not the bound)
#define RT_LAUNCH_PACKET(K) hipLaunchKernelGGL(K, grid, block, 0, s, S, pp, q, qg, t_min, t_max, hb, d_counters, \
                                               f.q_out, f.d_out_counts, f.d_path_radiance)
    if (fuse && gen) { if (stats) RT_LAUNCH_PACKET((k_trace_packet<true, true, true>)); else RT_LAUNCH_PACKET((k_trace_packet<false, true, true>)); }
    else if (gen)    { if (stats) RT_LAUNCH_PACKET((k_trace_packet<true, true, false>)); else RT_LAUNCH_PACKET((k_trace_packet<false, true, false>)); }
    else             { if (stats) RT_LAUNCH_PACKET((k_trace_packet<true, false, false>)); else RT_LAUNCH_PACKET((k_trace_packet<false, false, false>)); }
#undef RT_LAUNCH_PACKET
}

// n_upper: upper bound of the queue's entries; n_estimate: what the host expects (0 = no idea): sizes the one-wave-per-wave
// launch, the persistent k_shade_rest takes whatever lies beyond it.  d_hist: where the kernel reports the real count.
void launch_shade(const DScene& S, const PassParams& pp, uint32_t depth, RayQueue q_in, const QueueGeom& qg, uint32_t n_upper,
                  uint32_t n_estimate, HitBuf hb, RayQueue q_out, uint32_t* d_out_counts, double* d_path_radiance, uint32_t* d_hist,
                  hipStream_t s, uint32_t exact_waves) {
    if (!n_upper) return;
    if (exact_waves) {         // the caller knows the queue's logical waves (host-driven loop)
        hipLaunchKernelGGL(k_shade, dim3((unsigned)(((size_t)exact_waves * 64u + BLOCK - 1) / BLOCK)), dim3(BLOCK), 0, s, S, pp, depth, q_in, qg,
                           hb, q_out, d_out_counts, d_path_radiance, d_hist);
        return;
    }
    // logical waves: NSHARD * (waves of the longest shard) for a sharded queue — a shard holds at most shard_cap entries
    // and the whole queue at most n_upper
    const size_t bound = qg.count ? (size_t)NSHARD * ((std::min<size_t>(qg.shard_cap, n_upper) + 63u) / 64u) : ((size_t)n_upper + 63u) / 64u;
    size_t covered = bound;
    if (qg.count && n_estimate) {
        // shards fill evenly (each takes the survivors of every NSHARD-th input wave): longest shard ~ estimate / NSHARD
        const size_t per_shard = (size_t)n_estimate / NSHARD + (size_t)n_estimate / (NSHARD * 16) + 256;
        covered = std::min(bound, (size_t)NSHARD * ((per_shard + 63u) / 64u));
    }
    hipLaunchKernelGGL(k_shade, dim3((unsigned)((covered * 64u + BLOCK - 1) / BLOCK)), dim3(BLOCK), 0, s, S, pp, depth, q_in, qg, hb, q_out,
                       d_out_counts, d_path_radiance, d_hist);
    if (covered < bound) {
        const size_t waves = std::min<size_t>(bound - covered, 1024);
        hipLaunchKernelGGL(k_shade_rest, dim3((unsigned)((waves * 64u + BLOCK - 1) / BLOCK)), dim3(BLOCK), 0, s, S, pp, depth, q_in, qg, hb,
                           q_out, d_out_counts, d_path_radiance, (uint32_t)covered);
    }
}
void launch_accumulate(const PassParams& pp, const double* d_path_radiance, uint32_t n_local, double* d_rgb_sum,
                       uint32_t* d_zero_counts, uint32_t n_zero, hipStream_t s) {
    if (!pp.n_pixels || !n_local) return;
    if (pp.group == 16u && pp.tiled && n_local % 16u == 0 && pp.n_pixels % 4u == 0) {
        const size_t blocks = std::max<size_t>(pp.n_pixels / 4u, d_zero_counts ? (n_zero + 63u) / 64u : 0);
        hipLaunchKernelGGL(k_accumulate16, dim3((unsigned)blocks), dim3(64), 0, s, d_path_radiance, pp.n_pixels, pp.cam.width, pp.cam.height,
                           n_local, d_rgb_sum, d_zero_counts, n_zero);
        return;
    }
    hipLaunchKernelGGL(k_accumulate, dim3(grid_for(std::max<size_t>(pp.n_pixels, d_zero_counts ? n_zero : 0))), dim3(BLOCK), 0, s,
                       d_path_radiance, pp.n_pixels, pp.cam.width, pp.cam.height, pp.tiled, pp.group, n_local, d_rgb_sum,
                       d_zero_counts, n_zero);
}
void launch_resolve(const double* d_sum, double* d_mean, size_t n, double total_spp, hipStream_t s) {
    if (!n) return;
    hipLaunchKernelGGL(k_resolve, dim3(grid_for(n)), dim3(BLOCK), 0, s, d_sum, d_mean, n, total_spp);
}
void launch_narrow_f32(const double* d_in, float* d_out, size_t n, hipStream_t s) {
    if (!n) return;
    hipLaunchKernelGGL(k_narrow_f32, dim3(grid_for(n)), dim3(BLOCK), 0, s, d_in, d_out, n);
}
void launch_resolve_f32(const float* d_sum, double* d_mean, size_t n, double total_spp, hipStream_t s) {
    if (!n) return;
    hipLaunchKernelGGL(k_resolve_f32, dim3(grid_for(n)), dim3(BLOCK), 0, s, d_sum, d_mean, n, total_spp);
}
void launch_gamma8(const double* d_mean, size_t n, uint8_t* d_out, hipStream_t s) {
    if (!n) return;
    hipLaunchKernelGGL(k_gamma8, dim3(grid_for(n)), dim3(BLOCK), 0, s, d_mean, n, d_out);
}
void launch_blend_pass(const double* d_pass_sum, double* d_mean_frame, uint8_t* d_rgba8, size_t n_pixels, uint32_t n_samples,
                       uint64_t total_samples, bool quantise, hipStream_t s) {
    if (!n_pixels) return;
    hipLaunchKernelGGL(k_blend_pass, dim3(grid_for(n_pixels)), dim3(BLOCK), 0, s, d_pass_sum, d_mean_frame, d_rgba8, n_pixels,
                       (double)n_samples, (double)total_samples, quantise ? 1u : 0u);
}
void launch_pack_hits(HitBuf hb, const QueueGeom& qg, rt_hit* out, hipStream_t s) {
    if (!qg.n_direct) return;
    hipLaunchKernelGGL(k_pack_hits, dim3(grid_for(qg.n_direct)), dim3(BLOCK), 0, s, hb, qg, out);
}
void launch_pack_rays(RayQueue q, const QueueGeom& qg, rt_ray* out, hipStream_t s) {
    if (!qg.n_direct) return;
    hipLaunchKernelGGL(k_pack_rays, dim3(grid_for(qg.n_direct)), dim3(BLOCK), 0, s, q, qg, out);
}

}  // namespace rt
