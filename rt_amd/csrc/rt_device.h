// rt_device.h — device-side f64 arithmetic of the render hot path (gfx950).
//
// Every routine cites the reference lines whose arithmetic it reproduces; operation order follows
// nalgebra 0.33 as recalled in SURVEY.md App. B (dot = (x0*y0 + x1*y1) + x2*y2, normalize = v / |v|).
// The translation unit is compiled with -ffp-contract=off: Rust never fuses a*b+c, and the
// closest-hit (prim, t) contract is bit-exact.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "rt_layout.h"

namespace rt {

struct D3 { double x, y, z; };

__device__ __forceinline__ D3 mk(double x, double y, double z) { D3 r; r.x = x; r.y = y; r.z = z; return r; }
__device__ __forceinline__ D3 ld3(const double* p) { return mk(p[0], p[1], p[2]); }
__device__ __forceinline__ D3 add(D3 a, D3 b) { return mk(a.x + b.x, a.y + b.y, a.z + b.z); }
__device__ __forceinline__ D3 sub(D3 a, D3 b) { return mk(a.x - b.x, a.y - b.y, a.z - b.z); }
__device__ __forceinline__ D3 mul(D3 a, D3 b) { return mk(a.x * b.x, a.y * b.y, a.z * b.z); }
__device__ __forceinline__ D3 scale(D3 a, double s) { return mk(a.x * s, a.y * s, a.z * s); }
// a / s for the three components with ONE reciprocal: y = RN(1/s), q0 = RN(a*y), r = a - s*q0 (exact, one fma),
// q = RN(q0 + r*y).  With y within an ulp of 1/s, q0 + r*y is a/s to ~2^-104 relative, and a quotient of two doubles
// is never that close to a rounding boundary without being exact, so q is the correctly rounded a / s — bit for bit
// what three IEEE divisions give (it is the quotient step of the compiler's own division sequence; tests/native/
// div_shared_check.c compares the two over 3e8 cases incl. significands of all ones, zero numerators of either sign and
// negative divisors).  20 instead of 33 f64 instructions per vector; a shaded vertex does three of these.  The form needs
// the remainder not to underflow and y to be finite and non-zero: outside 2^-500 < |s| < 2^500 (never on the path: s is a
// vector length or a colour component) the plain divisions are used; numerators are finite and, other than 0, not below
// 2^-900 in magnitude (directions and colours).  A zero numerator would come out as +0 whatever its sign (the remainder
// step adds zeros of opposite sign), so the quotient's sign is set to sign(a) ^ sign(s) explicitly — two integer
// instructions per component — which is what the division gives for every operand pair.
__device__ __forceinline__ double quot_sign(double q, double a, double s) {
    const uint32_t sg = ((uint32_t)__double2hiint(a) ^ (uint32_t)__double2hiint(s)) & 0x80000000u;
    return __hiloint2double((int)(((uint32_t)__double2hiint(q) & 0x7fffffffu) | sg), __double2loint(q));
}
__device__ __forceinline__ D3 divs(D3 a, double s) {
    const double m = fabs(s);
    if (!(m > 0x1.0p-500 && m < 0x1.0p500)) return mk(a.x / s, a.y / s, a.z / s);
    const double y = 1.0 / s;
    const double qx = a.x * y, qy = a.y * y, qz = a.z * y;
    return mk(quot_sign(fma(fma(-s, qx, a.x), y, qx), a.x, s), quot_sign(fma(fma(-s, qy, a.y), y, qy), a.y, s),
              quot_sign(fma(fma(-s, qz, a.z), y, qz), a.z, s));
}
__device__ __forceinline__ D3 neg(D3 a) { return mk(-a.x, -a.y, -a.z); }
__device__ __forceinline__ double dot(D3 a, D3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }
__device__ __forceinline__ double norm2(D3 a) { return dot(a, a); }
__device__ __forceinline__ D3 cross(D3 a, D3 b) {
    return mk(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}
__device__ __forceinline__ D3 normalize(D3 a) { return divs(a, sqrt(norm2(a))); }

constexpr double F64_EPSILON = 2.220446049250313e-16;
constexpr uint32_t DEPTH_PRIMARY = 0xFFFFFFFFu;

// ---- Philox4x32-10 counter RNG ---------------------------------------
struct U4 { uint32_t x, y, z, w; };
__device__ __forceinline__ U4 philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        // one 32x32->64 multiply per product (v_mad_u64_u32): integer multiplies are quarter-rate on CDNA, and the
        // 120 mul_hi/mul_lo pairs of three blocks were a third of the camera-ray kernel's issue time
        const uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
        const uint32_t h0 = (uint32_t)(p0 >> 32), l0 = (uint32_t)p0, h1 = (uint32_t)(p1 >> 32), l1 = (uint32_t)p1;
        uint32_t n0 = h1 ^ c1 ^ k0, n2 = h0 ^ c3 ^ k1;
        c0 = n0; c1 = l1; c2 = n2; c3 = l0;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    U4 o; o.x = c0; o.y = c1; o.z = c2; o.w = c3;
    return o;
}
__device__ __forceinline__ double u64_to_unit(uint32_t lo, uint32_t hi) {
    uint64_t bits = ((uint64_t)hi << 32) | lo;
    return (double)(bits >> 11) * 0x1.0p-53;
}
struct RngKey { uint64_t seed; uint32_t pixel, sample, depth; };
// block b holds slots 2b (words x,y) and 2b+1 (words z,w)
__device__ __forceinline__ U4 rng_block(const RngKey& k, uint32_t block) {
    return philox4x32_10(k.pixel, k.sample, k.depth, block, (uint32_t)k.seed, (uint32_t)(k.seed >> 32));
}

// ---- Halton (src/camera.rs:140-165): element `index` = radical inverse of index+1, exact n/d ----
__device__ __forceinline__ double halton2(uint32_t index) {
    uint32_t i = index + 1;                       // <= 2^20
    int bits = 32 - __clz(i);
    uint32_t n = __brev(i) >> (32 - bits);
    return (double)n / (double)(1u << bits);
}
__device__ __forceinline__ double halton3(uint32_t index) {
    uint32_t i = index + 1, n = 0, d = 1;         // 3^13 > 2^20, fits in u32
    while (i) { n = n * 3u + i % 3u; i /= 3u; d *= 3u; }
    return (double)n / (double)d;
}

// ---- path slot -> pixel --------------------------------------------------------------------
// Slots enumerate the full 8x8 tiles first (tile-major, Z-order inside a tile: any aligned run of 4, 16
// or 64 slots is a 2x2, 4x4 or 8x8 pixel block), then the right strip, then the bottom strip.
__device__ __forceinline__ uint32_t compact3(uint32_t v) {      // bits 0, 2, 4 -> 0, 1, 2
    return (v & 1u) | ((v >> 1) & 2u) | ((v >> 2) & 4u);
}
__device__ __forceinline__ uint32_t slot_to_pixel(uint32_t slot, uint32_t W, uint32_t H, uint32_t tiled) {
    if (!tiled) return slot;
    const uint32_t W8 = W & ~7u, H8 = H & ~7u, nfull = W8 * H8;
    uint32_t x, y;
    if (slot < nfull) {
        const uint32_t tile = slot >> 6, in = slot & 63u, tpr = W8 >> 3;
        x = (tile % tpr) * 8u + compact3(in);
        y = (tile / tpr) * 8u + compact3(in >> 1);
    } else {
        uint32_t r = slot - nfull;
        const uint32_t wr = W - W8;
        if (r < wr * H8) {
            x = W8 + r % wr;
            y = r / wr;
        } else {
            r -= wr * H8;
            x = r % W;
            y = H8 + r / W;
        }
    }
    return y * W + x;
}

// ---- path id <-> (pixel slot, local sample) ----------------------------------------------------
// group G = samples of one pixel that share a wave of camera rays (1, 4, 16 or 64): a wave of 64 consecutive
// path ids holds G samples of each of 64/G neighbouring pixels.  The rays of one pixel differ only by the
// sub-pixel jitter, so the wave of the packet traversal walks (almost) one ray's nodes instead of the union
// over an 8x8 tile.  G == 1 is the plain sample-major order.  G > 1 needs n_pixels % (64/G) == 0 and a
// sample count per pass that is a multiple of G (the host checks).
__device__ __forceinline__ void pid_to_slot_sample(uint32_t pid, uint32_t n_pixels, uint32_t G, uint32_t& slot, uint32_t& local) {
    if (G <= 1u) {
        local = pid / n_pixels;
        slot = pid - local * n_pixels;
        return;
    }
    const uint32_t P = 64u / G, per_group = n_pixels * G;
    const uint32_t sg = pid / per_group, r = pid - sg * per_group;
    const uint32_t lane = r & 63u;
    slot = (r >> 6) * P + (lane & (P - 1u));
    local = sg * G + lane / P;
}
__device__ __forceinline__ size_t slot_sample_to_pid(uint32_t slot, uint32_t local, uint32_t n_pixels, uint32_t G) {
    if (G <= 1u) return (size_t)local * n_pixels + slot;
    const uint32_t P = 64u / G;
    return (size_t)(local / G) * ((size_t)n_pixels * G) + (size_t)(slot / P) * 64u + (local % G) * P + (slot % P);
}

// ---- path id -> (pixel, x, y, local sample) with the pass's precomputed divisors ------------------
__device__ __forceinline__ uint32_t fast_div(uint32_t x, uint32_t m, uint32_t s1, uint32_t s2) {
    const uint32_t t = __umulhi(m, x);
    return (t + ((x - t) >> s1)) >> s2;
}
struct PathIndex {
    uint32_t pixel, x, y, local;
};
// the same mapping as pid_to_slot_sample + slot_to_pixel (tested equal), one mul_hi division instead of five u32 divisions
template <class PP>
__device__ __forceinline__ PathIndex decode_path(const PP& pp, uint32_t pid) {
    PathIndex r;
    uint32_t slot;
    const uint32_t q = fast_div(pid, pp.div_group.m, pp.div_group.s1, pp.div_group.s2);       // pid / (n_pixels * G)
    const uint32_t rem = pid - q * pp.div_group.d;
    if (pp.group <= 1u) {
        r.local = q;
        slot = rem;
    } else {
        const uint32_t lane = rem & 63u, pm1 = (1u << pp.log2_p) - 1u;
        slot = ((rem >> 6) << pp.log2_p) + (lane & pm1);
        r.local = q * pp.group + (lane >> pp.log2_p);
    }
    const uint32_t W = pp.cam.width, H = pp.cam.height;
    if (!pp.tiled) {
        r.pixel = slot;
        r.y = fast_div(slot, pp.div_width.m, pp.div_width.s1, pp.div_width.s2);
        r.x = slot - r.y * W;
        return r;
    }
    const uint32_t W8 = W & ~7u, H8 = H & ~7u, nfull = W8 * H8;
    if (slot < nfull) {
        const uint32_t tile = slot >> 6, in = slot & 63u;
        const uint32_t ty = fast_div(tile, pp.div_tpr.m, pp.div_tpr.s1, pp.div_tpr.s2), tx = tile - ty * pp.div_tpr.d;
        r.x = tx * 8u + compact3(in);
        r.y = ty * 8u + compact3(in >> 1);
    } else {                                  // the strips beside / below the full tiles (sizes that are not multiples of 8)
        uint32_t k = slot - nfull;
        const uint32_t wr = W - W8;
        if (k < wr * H8) {
            r.x = W8 + k % wr;
            r.y = k / wr;
        } else {
            k -= wr * H8;
            r.x = k % W;
            r.y = H8 + k / W;
        }
    }
    r.pixel = r.y * W + r.x;
    return r;
}

// ---- sharded queue addressing ----------------------------------------------------------------
// A wave's view of a queue: the shard counts are read once per wave from device memory (lane k keeps shard k's) and
// looked up with v_readlane; a dense queue needs no memory at all.
struct WaveQueue {
    uint32_t cnt;          // lane k: live entries in shard k (sharded queues)
    uint32_t n_waves;      // logical size of the queue in waves (wave-uniform)
    uint32_t shard_cap, n_direct;
    bool direct;
    // lane = 0..63 of this wave; all 64 lanes call
    __device__ __forceinline__ void load(const QueueGeom& g, uint32_t lane) {
        shard_cap = g.shard_cap; n_direct = g.n_direct; direct = g.count == nullptr;
        cnt = 0;
        if (direct) { n_waves = (n_direct + 63u) >> 6; return; }
        cnt = g.count[COUNT_STRIDE * lane];
        uint32_t w = (cnt + 63u) >> 6;
        for (int off = 32; off; off >>= 1) w = max(w, (uint32_t)__shfl_xor((int)w, off));
        n_waves = (uint32_t)__builtin_amdgcn_readfirstlane((int)w) * NSHARD;
    }
    // One-shot waves (k_shade: a wave serves ONE logical wave) read two words through the scalar cache instead of the 64
    // counters: the queue's size in waves, left in count[QUEUE_NWAVES_WORD] by the traversal kernel that ran over the
    // same queue just before, and (in slot_light) their own shard's count.
    __device__ __forceinline__ void load_light(const QueueGeom& g) {
        shard_cap = g.shard_cap; n_direct = g.n_direct; direct = g.count == nullptr;
        cnt = 0;
        n_waves = direct ? (n_direct + 63u) >> 6 : g.count[QUEUE_NWAVES_WORD];
    }
    __device__ __forceinline__ bool slot_light(const QueueGeom& g, uint32_t wave, uint32_t lane, uint32_t& i) const {
        const uint32_t shard = wave & (NSHARD - 1);
        const uint32_t off = (wave / NSHARD) * 64u + lane;
        i = shard * shard_cap + off;
        if (direct) return wave * 64u + lane < n_direct;
        return wave < n_waves && off < g.count[COUNT_STRIDE * shard];
    }
    // wave: logical wave index (wave-uniform).  Returns whether this lane holds a live entry and its physical slot.
    __device__ __forceinline__ bool slot(uint32_t wave, uint32_t lane, uint32_t& i) const {
        const uint32_t shard = wave & (NSHARD - 1);
        const uint32_t off = (wave / NSHARD) * 64u + lane;
        i = shard * shard_cap + off;
        if (direct) return wave * 64u + lane < n_direct;
        return wave < n_waves && off < (uint32_t)__builtin_amdgcn_readlane((int)cnt, __builtin_amdgcn_readfirstlane((int)shard));
    }
};
// dense queues only (ray generation, packing): logical entry p -> physical slot
__device__ __forceinline__ bool direct_slot(const QueueGeom& g, uint32_t p, uint32_t& i) {
    const uint32_t wave = p >> 6;
    i = (wave & (NSHARD - 1)) * g.shard_cap + (wave / NSHARD) * 64u + (p & 63u);
    return p < g.n_direct;
}

// ---- primitives -----------------------------------------------------------------------------
// Triangle::hit (src/hittable.rs:408-495) on the precomputed edges; returns true and (dist,u,v).
__device__ __forceinline__ bool tri_test(D3 a, D3 e1, D3 e2, D3 o, D3 d, double lo, double hi,
                                         double& dist, double& bu, double& bv) {
    D3 uvec = cross(d, e2);
    double det = dot(e1, uvec);
    if (det < F64_EPSILON) return false;
    double inv = 1.0 / det;
    D3 ao = sub(o, a);
    double u = dot(ao, uvec) * inv;
    if (!(0.0 <= u && u <= 1.0)) return false;
    D3 vvec = cross(ao, e1);
    double v = dot(d, vvec) * inv;
    if (v < 0.0 || u + v > 1.0) return false;
    double t = dot(e2, vvec) * inv;
    if (!(lo <= t && t < hi)) return false;
    if (!(t > F64_EPSILON)) return false;
    dist = t; bu = u; bv = v;
    return true;
}

// tri_test with the caller's acceptance code run INSIDE the innermost success path (same operations in the same order).
// In the per-lane walk `if (tri_test(..., t, u, v)) accept(t, u, v)` left t, u, v undefined on the failing paths; the
// compiler carried them — and copies of the hit state — in registers across the whole traversal loop and moved them
// about on every iteration (a dozen v_mov per step).  With the use next to the definition there is nothing to merge.
template <class OnHit>
__device__ __forceinline__ void tri_test_do(D3 a, D3 e1, D3 e2, D3 o, D3 d, double lo, double hi, OnHit&& on_hit) {
    D3 uvec = cross(d, e2);
    double det = dot(e1, uvec);
    if (det < F64_EPSILON) return;
    double inv = 1.0 / det;
    D3 ao = sub(o, a);
    double u = dot(ao, uvec) * inv;
    if (!(0.0 <= u && u <= 1.0)) return;
    D3 vvec = cross(ao, e1);
    double v = dot(d, vvec) * inv;
    if (v < 0.0 || u + v > 1.0) return;
    double t = dot(e2, vvec) * inv;
    if (!(lo <= t && t < hi)) return;
    if (!(t > F64_EPSILON)) return;
    on_hit(t, u, v);
}

// Sphere::hit (src/hittable.rs:318-365) up to the NaN-uv rejection, which reduces to
// !(|z'| <= 1) for the rotated normal's z.
__device__ __forceinline__ bool sph_test(D3 c, double r, D3 rot2, D3 o, D3 d, double lo, double hi, double& tout) {
    D3 oc = sub(c, o);
    double a = norm2(d);
    double h = dot(d, oc);
    double cc = norm2(oc) - r * r;
    double disc = h * h - a * cc;
    if (disc < 0.0) return false;
    double sq = sqrt(disc);
    double t = (h - sq) / a;
    if (!(lo <= t && t < hi)) {
        t = (h + sq) / a;
        if (!(lo <= t && t < hi)) return false;
    }
    D3 p = add(scale(d, t), o);
    D3 n = divs(sub(p, c), r);
    if (!(dot(d, n) < 0.0)) n = neg(n);
    double zr = (rot2.x * n.x + rot2.y * n.y) + rot2.z * n.z;
    if (!(fabs(zr) <= 1.0)) return false;
    tout = t;
    return true;
}
// (the same with the acceptance code inside the success path: see tri_test_do)
template <class OnHit>
__device__ __forceinline__ void sph_test_do(D3 c, double r, D3 rot2, D3 o, D3 d, double lo, double hi, OnHit&& on_hit) {
    D3 oc = sub(c, o);
    double a = norm2(d);
    double h = dot(d, oc);
    double cc = norm2(oc) - r * r;
    double disc = h * h - a * cc;
    if (disc < 0.0) return;
    double sq = sqrt(disc);
    double t = (h - sq) / a;
    if (!(lo <= t && t < hi)) {
        t = (h + sq) / a;
        if (!(lo <= t && t < hi)) return;
    }
    D3 p = add(scale(d, t), o);
    D3 n = divs(sub(p, c), r);
    if (!(dot(d, n) < 0.0)) n = neg(n);
    double zr = (rot2.x * n.x + rot2.y * n.y) + rot2.z * n.z;
    if (!(fabs(zr) <= 1.0)) return;
    on_hit(t);
}

}  // namespace rt

// ---------------------------------------------------------------------------------------------
// shading-side helpers (used by the shade kernel only)
// ---------------------------------------------------------------------------------------------
namespace rt {

constexpr double PI_D = 3.14159265358979323846264338327950288;
constexpr double TAU_D = 6.28318530717958647692528676655900577;

// Rust `as i32` on f64: saturating, NaN -> 0 (src/texture.rs:63-65)
__device__ __forceinline__ int32_t f64_as_i32(double x) {
    if (x != x) return 0;
    if (x >= 2147483647.0) return 2147483647;
    if (x <= -2147483648.0) return (int32_t)0x80000000;
    return (int32_t)x;
}
__device__ __forceinline__ double clamp01(double x) {   // f64::clamp(0,1); NaN stays NaN
    if (x < 0.0) return 0.0;
    if (x > 1.0) return 1.0;
    return x;
}

// unit_sphere_uv (src/hittable.rs:370-388) + to_unit_spherical (:402-406) with the per-sphere
// rotation matrix precomputed at flatten time (unit_sphere_uv_facing, :392-400).
__device__ __forceinline__ void sphere_uv(D3 n, const double* r, double& u, double& v) {
    D3 q = mk((r[0] * n.x + r[1] * n.y) + r[2] * n.z,
              (r[3] * n.x + r[4] * n.y) + r[5] * n.z,
              (r[6] * n.x + r[7] * n.y) + r[8] * n.z);
    double theta = acos(-q.z);
    double phi = atan2(q.y, q.x) + PI_D;
    phi = phi + 0.0;
    double m = fmod(phi, TAU_D);
    if (m < 0.0) m += TAU_D;
    u = m / TAU_D;
    v = theta / PI_D;
}

struct SurfaceHit {
    D3 p, n;
    bool front;
    bool is_sphere;
    uint32_t material;
    // uv is evaluated lazily (only image textures read it)
    const double* bu;              // triangle barycentrics of this path's hit (HitBuf slots)
    const double* bv;
    const TriShade* ts;
    const SphShade* ss;
};

// Out of line on purpose: acos/atan2 in f64 are register-hungry and only image textures need uv;
// inlining them costs the whole shade kernel two waves per SIMD of occupancy.
// (arguments and result by value: a reference parameter would put the hit record on the scratch stack of every wave)
struct UV { double u, v; };
__device__ __noinline__ UV surface_uv(bool is_sphere, D3 n, const double* rot, const TriShade* ts, const double* bu, const double* bv) {
    UV r;
    if (is_sphere) {
        sphere_uv(n, rot, r.u, r.v);
    } else {                        // bbox-of-uvs mapping (src/hittable.rs:469-481)
        r.u = ts->uv_left + ts->uv_width * *bu;
        r.v = ts->uv_bot + ts->uv_height * *bv;
    }
    return r;
}

// Texture::value for the three kinds (src/texture.rs:25-29, :61-74, :99-118)
__device__ __forceinline__ D3 texture_value(const DScene& S, uint32_t ti, const SurfaceHit& h) {
    // (solid root textures never get here: k_shade takes their colour from the ShadeMat record)
    for (int guard = 0; guard < 64; ++guard) {
        const rt_texture* t = &S.texs[ti];
        uint32_t kind = t->kind;
        if (kind == RT_TEX_SOLID) return ld3(t->rgb);
        if (kind == RT_TEX_CHECKER) {
            double is = t->inv_scale;
            int32_t xi = f64_as_i32(floor(is * h.p.x));
            int32_t yi = f64_as_i32(floor(is * h.p.y));
            int32_t zi = f64_as_i32(floor(is * h.p.z));
            int32_t sum = (int32_t)((uint32_t)xi + (uint32_t)yi + (uint32_t)zi);
            ti = (sum % 2 == 0) ? t->even : t->odd;
            continue;
        }
        const UV uv = surface_uv(h.is_sphere, h.n, h.is_sphere ? h.ss->rot : nullptr, h.ts, h.bu, h.bv);
        const double u = uv.u, v = uv.v;
        double uc = clamp01(u), vc = clamp01(v);
        uint32_t w = t->width, ht = t->height;
        double fx = uc * (double)(w - 1), fy = vc * (double)(ht - 1);
        uint64_t x = (fx != fx || fx <= 0.0) ? 0ull : (uint64_t)fx;     // `as usize`: NaN/negative -> 0
        uint64_t y = (fy != fy || fy <= 0.0) ? 0ull : (uint64_t)fy;
        uint64_t texel = y * w + x;
        if (t->texel_format == RT_TEXEL_RGBA16) {
            const ushort4 px = *reinterpret_cast<const ushort4*>(S.texels + t->texel_offset + 8 * texel);
            return mk((double)px.x / 65535.0, (double)px.y / 65535.0, (double)px.z / 65535.0);
        }
        const uchar4 px = *reinterpret_cast<const uchar4*>(S.texels + t->texel_offset + 4 * texel);
        return mk((double)px.x / 255.0, (double)px.y / 255.0, (double)px.z / 255.0);
    }
    return mk(0.0, 0.0, 0.0);
}

// reflect / refract / Schlick (src/material.rs:46-60, :182-186)
__device__ __forceinline__ D3 reflect(D3 d, D3 n) { return sub(d, scale(scale(n, dot(d, n)), 2.0)); }
__device__ __forceinline__ D3 refract(D3 d, D3 n, double ratio) {
    double c = fmin(-dot(d, n), 1.0);
    D3 perp = scale(add(d, scale(n, c)), ratio);
    double x = -sqrt(fabs(1.0 - norm2(perp)));
    return add(scale(n, x), perp);
}
__device__ __forceinline__ double reflectance(double cosine, double ri) {
    double r0 = (1.0 - ri) / (1.0 + ri);
    r0 = r0 * r0;
    double a = 1.0 - cosine, a2 = a * a, a4 = a2 * a2;
    return r0 + (1.0 - r0) * (a * a4);          // powi(5)
}
__device__ __forceinline__ bool near_zero(D3 a) {   // src/vec3.rs:105-111
    const double e = 1.4901161193847656e-08;      // sqrt(f64::EPSILON), exact
    return fabs(a.x) < e && fabs(a.y) < e && fabs(a.z) < e;
}

// hw_skymodel::rgb::SkyState::radiance, f32 (recalled)
__device__ __forceinline__ float sky_radiance(const rt_sky& sky, float theta, float gamma, int ch) {
    const float* p = &sky.params[9 * ch];
    float cos_gamma = cosf(gamma);
    float cos_gamma2 = cos_gamma * cos_gamma;
    float cos_theta = fabsf(cosf(theta));
    float exp_m = expf(p[4] * gamma);
    float ray_m = cos_gamma2;
    float mie_m_lhs = 1.0f + cos_gamma2;
    // powf(b, 1.5): b * sqrt(b) is within 1.5 ulp of it (sqrt is correctly rounded; NaN for b < 0, 0 and inf kept) and a tenth
    // of OCML powf's ~150 instructions — the sky is evaluated once for nearly every path, three channels each
    const float mie_b = 1.0f + p[8] * p[8] - 2.0f * p[8] * cos_gamma;
    float mie_m_rhs = mie_b * sqrtf(mie_b);
    float mie_m = mie_m_lhs / mie_m_rhs;
    float zenith = sqrtf(cos_theta);
    float radiance_lhs = 1.0f + p[0] * expf(p[1] / (cos_theta + 0.01f));
    float radiance_rhs = p[2] + p[3] * exp_m + p[5] * ray_m + p[6] * mie_m + p[7] * zenith;
    float radiance_dist = radiance_lhs * radiance_rhs;
    return sky.radiances[ch] * radiance_dist;
}
__device__ __forceinline__ double tonemap1(double x) {   // uncharted2_tonemap (src/hittable.rs:51-66)
    const double a = 0.15, b = 0.50, c = 0.10, d = 0.20, e = 0.02, f = 0.30;
    double num = x * (a * x + c * b) + d * e;
    double den = x * (a * x + b) + d * f;
    return num / den - e / f;
}
// World::sky_color_toward (src/hittable.rs:84-93) + uncharted2 (:70-79)
__device__ __forceinline__ D3 sky_color_toward(const rt_sky& sky, D3 dir) {
    float theta = (float)acos(dir.z);
    double g = dot(dir, ld3(sky.sun));
    g = g < -1.0 ? -1.0 : (g > 1.0 ? 1.0 : g);
    float gamma = (float)g;
    double cr = (double)sky_radiance(sky, theta, gamma, 0);
    double cg = (double)sky_radiance(sky, theta, gamma, 1);
    double cb = (double)sky_radiance(sky, theta, gamma, 2);
    const double exposure_bias = 1.1, w = 11.2;
    double ws = 1.0 / tonemap1(w);
    return mk(ws * tonemap1(exposure_bias * cr), ws * tonemap1(exposure_bias * cg), ws * tonemap1(exposure_bias * cb));
}

}  // namespace rt
