/* rt_oracle.c — CPU restatement (plain C11, f64) of thabnir/rt's render hot path.
 *
 * TEST INFRASTRUCTURE ONLY — see rt_oracle.h.  PARITY UNPINNED against the real Rust binary (no
 * toolchain, no reference fixtures); pinned by closed-form KATs derived from the source text.
 *
 * Every function cites the reference lines it follows (paths relative to /root/reference).
 * Compile with -ffp-contract=off: Rust never contracts a*b+c into an FMA, so neither may we.
 * Operation order of the vector helpers follows nalgebra 0.33 as recalled in SURVEY.md App. B.
 */
#define _GNU_SOURCE
#include "rt_oracle.h"

#include <float.h>
#include <math.h>
#include <pthread.h>
#include <stdatomic.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------------------------------ */
/* nalgebra Vector3<f64> (src/vec3.rs:7-10)                                                    */
/* ------------------------------------------------------------------------------------------ */
typedef struct { double x, y, z; } v3;

static inline v3 V(double x, double y, double z) { v3 r = {x, y, z}; return r; }
static inline v3 vadd(v3 a, v3 b) { return V(a.x + b.x, a.y + b.y, a.z + b.z); }
static inline v3 vsub(v3 a, v3 b) { return V(a.x - b.x, a.y - b.y, a.z - b.z); }
static inline v3 vmul(v3 a, v3 b) { return V(a.x * b.x, a.y * b.y, a.z * b.z); }   /* component_mul */
static inline v3 vscale(v3 a, double s) { return V(a.x * s, a.y * s, a.z * s); }
static inline v3 vdivs(v3 a, double s) { return V(a.x / s, a.y / s, a.z / s); }
static inline v3 vneg(v3 a) { return V(-a.x, -a.y, -a.z); }
/* nalgebra's unrolled 3-vector dot: (x0*y0 + x1*y1) + x2*y2 */
static inline double vdot(v3 a, v3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }
static inline double vnorm2(v3 a) { return vdot(a, a); }
static inline v3 vcross(v3 a, v3 b) {
    return V(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}
/* normalize = self / self.norm(): per-component division */
static inline v3 vnormalize(v3 a) { return vdivs(a, sqrt(vnorm2(a))); }
static inline v3 from3(const double* p) { return V(p[0], p[1], p[2]); }
static inline void to3(v3 a, double* p) { p[0] = a.x; p[1] = a.y; p[2] = a.z; }

/* bvh::ray::Ray<f64,3> (src/vec3.rs:6): Ray::new normalises the direction, stores 1/d. */
typedef struct { v3 o, d, inv; } ray_t;
static inline ray_t ray_new(v3 o, v3 dir) {
    ray_t r;
    r.o = o;
    r.d = vnormalize(dir);
    r.inv = V(1.0 / r.d.x, 1.0 / r.d.y, 1.0 / r.d.z);
    return r;
}

/* ------------------------------------------------------------------------------------------ */
/* Counter-based RNG contract: Philox4x32-10                            */
/* ------------------------------------------------------------------------------------------ */
void orc_philox4x32_10(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]) {
    uint32_t c0 = ctr[0], c1 = ctr[1], c2 = ctr[2], c3 = ctr[3];
    uint32_t k0 = key[0], k1 = key[1];
    for (int r = 0; r < 10; ++r) {
        uint64_t p0 = (uint64_t)0xD2511F53u * c0;
        uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
        uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
        uint32_t n1 = (uint32_t)p1;
        uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
        uint32_t n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

#define DEPTH_PRIMARY 0xFFFFFFFFu

/* Uniform f64 in [0,1): slot s lives in Philox block s>>1, words 2*(s&1) (lo) and +1 (hi). */
double orc_uniform(uint64_t seed, uint32_t pixel, uint32_t sample, uint32_t depth, uint32_t slot) {
    uint32_t ctr[4] = {pixel, sample, depth, slot >> 1};
    uint32_t key[2] = {(uint32_t)seed, (uint32_t)(seed >> 32)};
    uint32_t o[4];
    orc_philox4x32_10(ctr, key, o);
    uint32_t w = (slot & 1u) * 2u;
    uint64_t bits = ((uint64_t)o[w + 1] << 32) | o[w];
    return (double)(bits >> 11) * 0x1.0p-53;
}

/* Slot table of a path vertex (depth = bounce index).  SURVEY.md App. C with slots 0 and 4 exchanged, so that the four
 * draws of a Lambertian vertex (roulette + random_unit) sit in two Philox blocks instead of three; the slots are
 * independent uniforms, so nothing else changes.  integration/rust/patches/rt-counter-rng.patch uses the same table. */
#define SLOT_ROULETTE 0u      /* russian_roulette: survive iff u < p        (src/camera.rs:288) */
#define SLOT_UNIT 1u          /* 1, 2, 3: Vec3::random_unit                 (src/vec3.rs:113-120) */
#define SLOT_NOISE 4u         /* Dielectric reflect-or-refract noise       (src/material.rs:164) */
typedef struct { uint64_t seed; uint32_t pixel, sample, depth; } rng_key;
static inline double rng_u(const rng_key* k, uint32_t slot) {
    return orc_uniform(k->seed, k->pixel, k->sample, k->depth, slot);
}
/* Vec3::random(-1,1).normalize() (src/vec3.rs:113-120) on slots 1,2,3 */
static inline v3 random_unit(const rng_key* k) {
    v3 r = V(2.0 * rng_u(k, SLOT_UNIT) - 1.0, 2.0 * rng_u(k, SLOT_UNIT + 1) - 1.0, 2.0 * rng_u(k, SLOT_UNIT + 2) - 1.0);
    return vnormalize(r);
}

/* ------------------------------------------------------------------------------------------ */
/* Halton (src/camera.rs:140-165)                                                              */
/* ------------------------------------------------------------------------------------------ */
static void halton_fill(uint64_t base, uint64_t len, double* out) {
    uint64_t n = 0, d = 1;
    for (uint64_t i = 0; i < len; ++i) {
        uint64_t x = d - n;
        if (x == 1) {
            n = 1;
            d *= base;
        } else {
            uint64_t y = d / base;
            while (x < y) y /= base;
            n = (base + 1) * y - x;
        }
        out[i] = (double)n / (double)d;
    }
}
double orc_halton_recurrence(uint64_t base, uint64_t index) {
    double* t = (double*)malloc((index + 1) * sizeof(double));
    if (!t) return NAN;
    halton_fill(base, index + 1, t);
    double r = t[index];
    free(t);
    return r;
}
/* closed form used by the kernels: radical inverse of (index+1) as the exact rational n/d */
double orc_halton_closed(uint32_t base, uint32_t index) {
    uint64_t i = (uint64_t)index + 1, n = 0, d = 1;
    while (i) { n = n * base + i % base; i /= base; d *= base; }
    return (double)n / (double)d;
}

/* ------------------------------------------------------------------------------------------ */
/* Scene                                                                                        */
/* ------------------------------------------------------------------------------------------ */
typedef struct { v3 lo, hi; uint32_t left, right, prim; } onode;   /* leaf: left == UINT32_MAX */

struct orc_scene {
    rt_sphere* sph; size_t nsph;
    rt_triangle* tri; size_t ntri;
    rt_material* mat; size_t nmat;
    rt_texture* tex; size_t ntex;
    uint8_t* texels; size_t texel_bytes;
    rt_sky sky;
    double (*sph_rot)[9];     /* per-sphere texture rotation (src/hittable.rs:376-377, :393-396) */
    onode* nodes; size_t nnodes;
};

typedef struct {
    v3 point, normal;
    double t, u, v;
    uint32_t prim, material;
    int front;
} isect;

/* Rotation3::from_euler_angles(roll, pitch, yaw) = Rz(yaw)*Ry(pitch)*Rx(roll) (nalgebra, recalled) */
static void euler(double roll, double pitch, double yaw, double m[9]) {
    double sr = sin(roll), cr = cos(roll), sp = sin(pitch), cp = cos(pitch), sy = sin(yaw), cy = cos(yaw);
    m[0] = cy * cp; m[1] = cy * sp * sr - sy * cr; m[2] = cy * sp * cr + sy * sr;
    m[3] = sy * cp; m[4] = sy * sp * sr + cy * cr; m[5] = sy * sp * cr - cy * sr;
    m[6] = -sp;     m[7] = cp * sr;                m[8] = cp * cr;
}
static void mat3mul(const double a[9], const double b[9], double c[9]) {
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j)
            c[3 * i + j] = (a[3 * i] * b[j] + a[3 * i + 1] * b[3 + j]) + a[3 * i + 2] * b[6 + j];
}
/* unit_sphere_uv_facing (src/hittable.rs:392-400): pitch/yaw of the facing, rotation matrix of
 * unit_sphere_uv (:376-377) */
static void sphere_rotation(const double f[3], double r[9]) {
    double pitch = atan2(f[2], sqrt(f[1] * f[1] + f[0] * f[0]));
    double yaw = atan2(f[1], f[0]);
    double a[9], b[9];
    euler(0.0, pitch, 0.0, a);
    euler(0.0, 0.0, -yaw, b);
    mat3mul(a, b, r);
}
static const double PI_ = 3.14159265358979323846264338327950288;
static const double TAU_ = 6.28318530717958647692528676655900577;

/* unit_sphere_uv (src/hittable.rs:370-388) + to_unit_spherical (:402-406) */
static void sphere_uv_rot(v3 n, const double r[9], double* u, double* v) {
    v3 q = V((r[0] * n.x + r[1] * n.y) + r[2] * n.z,
             (r[3] * n.x + r[4] * n.y) + r[5] * n.z,
             (r[6] * n.x + r[7] * n.y) + r[8] * n.z);
    double theta = acos(-q.z);
    double phi = atan2(q.y, q.x) + PI_;
    phi = phi + 0.0;                       /* rotation_rads = 0.0 (:398) */
    double m = fmod(phi, TAU_);            /* rem_euclid(TAU) */
    if (m < 0.0) m += TAU_;
    *u = m / TAU_;
    *v = theta / PI_;
}
void orc_sphere_uv(const double n[3], const double face[3], double uv[2]) {
    double r[9];
    sphere_rotation(face, r);
    sphere_uv_rot(from3(n), r, &uv[0], &uv[1]);
}

/* Sphere::hit (src/hittable.rs:318-365) */
static int sphere_hit(const rt_sphere* s, const double rot[9], uint32_t prim, const ray_t* r,
                      double lo, double hi, isect* out) {
    v3 c = from3(s->center);
    v3 oc = vsub(c, r->o);
    double a = vnorm2(r->d);
    double h = vdot(r->d, oc);
    double cc = vnorm2(oc) - s->radius * s->radius;
    double disc = h * h - a * cc;
    if (disc < 0.0) return 0;
    double sq = sqrt(disc);
    double t = (h - sq) / a;
    if (!(lo <= t && t < hi)) {
        t = (h + sq) / a;
        if (!(lo <= t && t < hi)) return 0;
    }
    v3 p = vadd(vscale(r->d, t), r->o);                 /* RayExt::at (src/vec3.rs:17-19) */
    v3 n = vdivs(vsub(p, c), s->radius);
    int front = vdot(r->d, n) < 0.0;                    /* Intersection::is_front_face (src/intersection.rs:36-38) */
    if (!front) n = vneg(n);
    double u, v;
    sphere_uv_rot(n, rot, &u, &v);
    if (isnan(u) || isnan(v)) return 0;                 /* :350-354 */
    out->point = p; out->normal = n; out->t = t; out->u = u; out->v = v;
    out->prim = prim; out->material = s->material; out->front = front;
    return 1;
}

static inline double min3(double a, double b, double c) { double m = a < b ? a : b; return m < c ? m : c; }
static inline double max3(double a, double b, double c) { double m = a > b ? a : b; return m > c ? m : c; }

/* Triangle::hit (src/hittable.rs:408-495) */
static int triangle_hit(const rt_triangle* tr, uint32_t prim, const ray_t* r, double lo, double hi, isect* out) {
    v3 a = from3(tr->a), b = from3(tr->b), c = from3(tr->c);
    v3 ab = vsub(b, a), ac = vsub(c, a);
    v3 uvec = vcross(r->d, ac);
    double det = vdot(ab, uvec);
    if (det < DBL_EPSILON) return 0;                    /* one-sided (:428) */
    double inv = 1.0 / det;
    v3 ao = vsub(r->o, a);
    double u = vdot(ao, uvec) * inv;
    if (!(0.0 <= u && u <= 1.0)) return 0;
    v3 vvec = vcross(ao, ab);
    double v = vdot(r->d, vvec) * inv;
    if (v < 0.0 || u + v > 1.0) return 0;
    double dist = vdot(ac, vvec) * inv;
    if (!(lo <= dist && dist < hi)) return 0;
    if (!(dist > DBL_EPSILON)) return 0;
    v3 p = vadd(r->o, vscale(r->d, dist));              /* origin first (:463) */
    v3 n = from3(tr->n);
    int front = vdot(r->d, n) <= 0.0;                   /* :464 */
    /* f64::min/max ignore NaN like fmin/fmax; inputs are finite */
    double left = fmin(fmin(tr->uv[0][0], tr->uv[1][0]), tr->uv[2][0]);
    double right = fmax(fmax(tr->uv[0][0], tr->uv[1][0]), tr->uv[2][0]);
    double bot = fmin(fmin(tr->uv[0][1], tr->uv[1][1]), tr->uv[2][1]);
    double top = fmax(fmax(tr->uv[0][1], tr->uv[1][1]), tr->uv[2][1]);
    double width = right - left, height = top - bot;
    out->point = p; out->normal = n; out->t = dist;
    out->u = left + width * u; out->v = bot + height * v;
    out->prim = prim; out->material = tr->material; out->front = front;
    return 1;
}

static inline int prim_hit(const orc_scene* S, uint32_t prim, const ray_t* r, double lo, double hi, isect* out) {
    if (prim < S->nsph) return sphere_hit(&S->sph[prim], S->sph_rot[prim], prim, r, lo, hi, out);
    return triangle_hit(&S->tri[prim - S->nsph], prim, r, lo, hi, out);
}

/* ------------------------------------------------------------------------------------------ */
/* BVH: binned SAH (6 buckets), ONE primitive per leaf, near-first traversal — mimics what      */
/* bvh 0.10 builds/does (recalled; topology does not influence results, ties excepted).        */
/* ------------------------------------------------------------------------------------------ */
typedef struct { v3 lo, hi, c; } pbox;

static inline v3 vmin(v3 a, v3 b) { return V(fmin(a.x, b.x), fmin(a.y, b.y), fmin(a.z, b.z)); }
static inline v3 vmax(v3 a, v3 b) { return V(fmax(a.x, b.x), fmax(a.y, b.y), fmax(a.z, b.z)); }
static inline double axis_of(v3 a, int k) { return k == 0 ? a.x : (k == 1 ? a.y : a.z); }
static inline double half_area(v3 lo, v3 hi) {
    v3 e = vsub(hi, lo);
    return e.x * e.y + e.y * e.z + e.z * e.x;
}

typedef struct { onode* nodes; size_t n, cap; const pbox* pb; uint32_t* idx; } builder;

static uint32_t build_rec(builder* B, size_t first, size_t count) {
    uint32_t me = (uint32_t)B->n++;
    v3 lo = V(INFINITY, INFINITY, INFINITY), hi = V(-INFINITY, -INFINITY, -INFINITY);
    v3 clo = lo, chi = hi;
    for (size_t i = first; i < first + count; ++i) {
        const pbox* p = &B->pb[B->idx[i]];
        lo = vmin(lo, p->lo); hi = vmax(hi, p->hi);
        clo = vmin(clo, p->c); chi = vmax(chi, p->c);
    }
    B->nodes[me].lo = lo; B->nodes[me].hi = hi;
    if (count == 1) {
        B->nodes[me].left = UINT32_MAX; B->nodes[me].right = UINT32_MAX;
        B->nodes[me].prim = B->idx[first];
        return me;
    }
    v3 ext = vsub(chi, clo);
    int ax = 0;
    if (ext.y > ext.x) ax = 1;
    if (ext.z > axis_of(ext, ax)) ax = 2;
    size_t mid = first + count / 2;
    double e = axis_of(ext, ax);
    if (e > 0.0) {
        enum { NB = 6 };
        size_t cnt[NB] = {0};
        v3 blo[NB], bhi[NB];
        for (int k = 0; k < NB; ++k) { blo[k] = V(INFINITY, INFINITY, INFINITY); bhi[k] = V(-INFINITY, -INFINITY, -INFINITY); }
        double c0 = axis_of(clo, ax);
        for (size_t i = first; i < first + count; ++i) {
            const pbox* p = &B->pb[B->idx[i]];
            int k = (int)((axis_of(p->c, ax) - c0) / e * NB);
            if (k >= NB) k = NB - 1;
            if (k < 0) k = 0;
            cnt[k]++; blo[k] = vmin(blo[k], p->lo); bhi[k] = vmax(bhi[k], p->hi);
        }
        double best = INFINITY; int bestk = -1;
        for (int s = 1; s < NB; ++s) {
            v3 llo = V(INFINITY, INFINITY, INFINITY), lhi = V(-INFINITY, -INFINITY, -INFINITY), rlo = llo, rhi = lhi;
            size_t lc = 0, rc = 0;
            for (int k = 0; k < s; ++k) if (cnt[k]) { lc += cnt[k]; llo = vmin(llo, blo[k]); lhi = vmax(lhi, bhi[k]); }
            for (int k = s; k < NB; ++k) if (cnt[k]) { rc += cnt[k]; rlo = vmin(rlo, blo[k]); rhi = vmax(rhi, bhi[k]); }
            if (!lc || !rc) continue;
            double cost = half_area(llo, lhi) * (double)lc + half_area(rlo, rhi) * (double)rc;
            if (cost < best) { best = cost; bestk = s; }
        }
        if (bestk > 0) {
            size_t i = first, j = first + count;
            while (i < j) {
                const pbox* p = &B->pb[B->idx[i]];
                int k = (int)((axis_of(p->c, ax) - c0) / e * NB);
                if (k >= NB) k = NB - 1;
                if (k < 0) k = 0;
                if (k < bestk) ++i;
                else { --j; uint32_t t = B->idx[i]; B->idx[i] = B->idx[j]; B->idx[j] = t; }
            }
            if (i > first && i < first + count) mid = i;
        }
    }
    uint32_t l = build_rec(B, first, mid - first);
    uint32_t r = build_rec(B, mid, first + count - mid);
    B->nodes[me].left = l; B->nodes[me].right = r; B->nodes[me].prim = UINT32_MAX;
    return me;
}

static int build_bvh(orc_scene* S) {
    size_t n = S->nsph + S->ntri;
    S->nodes = NULL; S->nnodes = 0;
    if (n == 0) return 0;
    pbox* pb = (pbox*)malloc(n * sizeof(pbox));
    uint32_t* idx = (uint32_t*)malloc(n * sizeof(uint32_t));
    onode* nodes = (onode*)malloc((2 * n) * sizeof(onode));
    if (!pb || !idx || !nodes) { free(pb); free(idx); free(nodes); return -1; }
    for (size_t i = 0; i < S->nsph; ++i) {               /* Sphere aabb (src/hittable.rs:299-306) */
        v3 c = from3(S->sph[i].center); double r = S->sph[i].radius;
        pb[i].lo = vsub(c, V(r, r, r)); pb[i].hi = vadd(c, V(r, r, r));
    }
    for (size_t i = 0; i < S->ntri; ++i) {               /* Triangle aabb (:242-248) */
        v3 a = from3(S->tri[i].a), b = from3(S->tri[i].b), c = from3(S->tri[i].c);
        pb[S->nsph + i].lo = vmin(vmin(a, b), c); pb[S->nsph + i].hi = vmax(vmax(a, b), c);
    }
    for (size_t i = 0; i < n; ++i) {
        pb[i].c = vscale(vadd(pb[i].lo, pb[i].hi), 0.5);
        idx[i] = (uint32_t)i;
    }
    builder B = {nodes, 0, 2 * n, pb, idx};
    build_rec(&B, 0, n);
    S->nodes = nodes; S->nnodes = B.n;
    free(pb); free(idx);
    return 0;
}

/* slab test with the ray's reciprocal direction; fmin/fmax drop NaNs (0*inf) conservatively, and
 * the interval is widened by a few ulps so a primitive touching its own box is never culled. */
static inline int box_hit(const onode* nd, const ray_t* r, double* t_enter) {
    double t1 = (nd->lo.x - r->o.x) * r->inv.x, t2 = (nd->hi.x - r->o.x) * r->inv.x;
    double tn = fmin(t1, t2), tf = fmax(t1, t2);
    t1 = (nd->lo.y - r->o.y) * r->inv.y; t2 = (nd->hi.y - r->o.y) * r->inv.y;
    tn = fmax(tn, fmin(t1, t2)); tf = fmin(tf, fmax(t1, t2));
    t1 = (nd->lo.z - r->o.z) * r->inv.z; t2 = (nd->hi.z - r->o.z) * r->inv.z;
    tn = fmax(tn, fmin(t1, t2)); tf = fmin(tf, fmax(t1, t2));
    tn -= fabs(tn) * 8e-16; tf += fabs(tf) * 8e-16;
    *t_enter = tn;
    return tf >= (tn > 0.0 ? tn : 0.0);
}

/* World::hit (src/hittable.rs:135-149): candidates near->far; each tested with [lo, nearest). */
static int world_hit(const orc_scene* S, const ray_t* r, double lo, double hi, int mode, isect* best) {
    double nearest = hi;
    int found = 0;
    size_t n = S->nsph + S->ntri;
    isect tmp;
    if (mode == ORC_BRUTE || S->nnodes == 0) {
        for (size_t i = 0; i < n; ++i)
            if (prim_hit(S, (uint32_t)i, r, lo, nearest, &tmp)) { nearest = tmp.t; *best = tmp; found = 1; }
        return found;
    }
    uint32_t stack[256]; int sp = 0;
    double te;
    if (!box_hit(&S->nodes[0], r, &te)) return 0;
    stack[sp++] = 0;
    while (sp) {
        const onode* nd = &S->nodes[stack[--sp]];
        if (nd->left == UINT32_MAX) {
            if (prim_hit(S, nd->prim, r, lo, nearest, &tmp)) { nearest = tmp.t; *best = tmp; found = 1; }
            continue;
        }
        double tl, tr;
        int hl = box_hit(&S->nodes[nd->left], r, &tl), hr = box_hit(&S->nodes[nd->right], r, &tr);
        if (mode == ORC_BVH_PRUNE) {                    /* not in the reference: bvh 0.10's iterator cannot see `nearest` */
            if (hl && tl > nearest) hl = 0;
            if (hr && tr > nearest) hr = 0;
        }
        if (sp > 250) return -1;
        if (hl && hr) {
            if (tl <= tr) { stack[sp++] = nd->right; stack[sp++] = nd->left; }
            else { stack[sp++] = nd->left; stack[sp++] = nd->right; }
        } else if (hl) stack[sp++] = nd->left;
        else if (hr) stack[sp++] = nd->right;
    }
    return found;
}

/* ------------------------------------------------------------------------------------------ */
/* Textures (src/texture.rs)                                                                    */
/* ------------------------------------------------------------------------------------------ */
static inline int32_t f64_as_i32(double x) {             /* Rust `as i32`: saturating, NaN -> 0 */
    if (isnan(x)) return 0;
    if (x >= 2147483647.0) return INT32_MAX;
    if (x <= -2147483648.0) return INT32_MIN;
    return (int32_t)x;
}
static inline uint64_t f64_as_usize(double x) {          /* Rust `as usize`: saturating, NaN -> 0 */
    if (isnan(x) || x <= 0.0) return 0;
    if (x >= 18446744073709551615.0) return UINT64_MAX;
    return (uint64_t)x;
}
static inline double clamp01(double x) {                 /* f64::clamp(0,1): NaN stays NaN */
    if (x < 0.0) return 0.0;
    if (x > 1.0) return 1.0;
    return x;
}
static v3 texture_value(const orc_scene* S, uint32_t ti, double u, double v, v3 p) {
    for (;;) {
        const rt_texture* t = &S->tex[ti];
        if (t->kind == RT_TEX_SOLID) return from3(t->rgb);          /* :25-29 */
        if (t->kind == RT_TEX_CHECKER) {                             /* :61-74 */
            int32_t xi = f64_as_i32(floor(t->inv_scale * p.x));
            int32_t yi = f64_as_i32(floor(t->inv_scale * p.y));
            int32_t zi = f64_as_i32(floor(t->inv_scale * p.z));
            int32_t sum = (int32_t)((uint32_t)xi + (uint32_t)yi + (uint32_t)zi);   /* release-mode wrap */
            ti = (sum % 2 == 0) ? t->even : t->odd;
            continue;
        }
        /* ImageTexture::value (:99-118); Image index (src/camera.rs:127-134) */
        double uc = clamp01(u), vc = clamp01(v);
        uint64_t x = f64_as_usize(uc * (double)(t->width - 1));
        uint64_t y = f64_as_usize(vc * (double)(t->height - 1));
        uint64_t texel = y * t->width + x;
        if (t->texel_format == RT_TEXEL_RGBA16) {
            const uint16_t* px = (const uint16_t*)(S->texels + t->texel_offset) + 4 * texel;
            return V((double)px[0] / 65535.0, (double)px[1] / 65535.0, (double)px[2] / 65535.0);
        }
        const uint8_t* px = S->texels + t->texel_offset + 4 * texel;  /* src/camera.rs:68-70 */
        return V((double)px[0] / 255.0, (double)px[1] / 255.0, (double)px[2] / 255.0);
    }
}
void orc_texture_value(const orc_scene* s, uint32_t tex, double u, double v, const double p[3], double out[3]) {
    to3(texture_value(s, tex, u, v, from3(p)), out);
}

/* ------------------------------------------------------------------------------------------ */
/* Materials (src/material.rs)                                                                  */
/* ------------------------------------------------------------------------------------------ */
static inline v3 reflect(v3 d, v3 n) {                   /* :46-51 */
    v3 scaled = vscale(n, vdot(d, n));
    return vsub(d, vscale(scaled, 2.0));
}
static inline v3 refract(v3 d, v3 n, double ratio) {      /* :54-60 */
    double c = fmin(-vdot(d, n), 1.0);
    v3 perp = vscale(vadd(d, vscale(n, c)), ratio);
    double x = -sqrt(fabs(1.0 - vnorm2(perp)));
    return vadd(vscale(n, x), perp);
}
static inline double powi5(double a) {                   /* compiler-rt __powidf2(a,5) */
    double a2 = a * a, a4 = a2 * a2;
    return a * a4;
}
static inline double reflectance(double cosine, double ri) {   /* :182-186 */
    double r0 = (1.0 - ri) / (1.0 + ri);
    r0 = r0 * r0;
    return r0 + (1.0 - r0) * powi5(1.0 - cosine);
}
void orc_reflect(const double d[3], const double n[3], double out[3]) { to3(reflect(from3(d), from3(n)), out); }
void orc_refract(const double d[3], const double n[3], double ratio, double out[3]) { to3(refract(from3(d), from3(n), ratio), out); }
double orc_reflectance(double cosine, double ri) { return reflectance(cosine, ri); }

static inline int near_zero(v3 a) {                      /* src/vec3.rs:105-111 */
    double e = sqrt(DBL_EPSILON);
    return fabs(a.x) < e && fabs(a.y) < e && fabs(a.z) < e;
}

/* Scatter::scatter for the three materials; never None (src/material.rs:40-43). */
static void scatter(const orc_scene* S, const ray_t* rin, const isect* h, const rng_key* k, v3* att, ray_t* out) {
    const rt_material* m = &S->mat[h->material];
    if (m->kind == RT_MAT_LAMBERTIAN) {                  /* :110-120 */
        v3 dir = vadd(h->normal, random_unit(k));
        if (near_zero(dir)) dir = h->normal;
        *out = ray_new(h->point, dir);
        *att = texture_value(S, m->texture, h->u, h->v, h->point);
    } else if (m->kind == RT_MAT_METAL) {                /* :94-108 */
        v3 dir = reflect(rin->d, h->normal);
        if (m->has_fuzz) dir = vadd(dir, vscale(random_unit(k), m->fuzz));
        *out = ray_new(h->point, dir);
        *att = texture_value(S, m->texture, h->u, h->v, h->point);
    } else {                                             /* Dielectric :150-179 */
        double ri = h->front ? 1.0 / m->ior : m->ior;
        v3 din = vnormalize(rin->d);
        double c = fmin(-vdot(din, h->normal), 1.0);
        double s = sqrt(1.0 - c * c);
        int cannot = ri * s > 1.0;
        double noise = rng_u(k, SLOT_NOISE);
        v3 dir;
        if (cannot || reflectance(c, ri) > noise) dir = reflect(din, h->normal);
        else if (m->has_fuzz) dir = vadd(refract(din, h->normal, ri), vscale(random_unit(k), m->fuzz));
        else dir = refract(din, h->normal, ri);
        *att = V(1.0, 1.0, 1.0);
        *out = ray_new(h->point, vnormalize(dir));
    }
}

/* ------------------------------------------------------------------------------------------ */
/* Sky + tonemap (src/hittable.rs:48-93); hw-skymodel radiance() recalled    */
/* ------------------------------------------------------------------------------------------ */
float orc_sky_radiance(const rt_sky* sky, float theta, float gamma, int ch) {
    const float* p = &sky->params[9 * ch];
    float r = sky->radiances[ch];
    float cos_gamma = cosf(gamma);
    float cos_gamma2 = cos_gamma * cos_gamma;
    float cos_theta = fabsf(cosf(theta));
    float exp_m = expf(p[4] * gamma);
    float ray_m = cos_gamma2;
    float mie_m_lhs = 1.0f + cos_gamma2;
    float mie_m_rhs = powf(1.0f + p[8] * p[8] - 2.0f * p[8] * cos_gamma, 1.5f);
    float mie_m = mie_m_lhs / mie_m_rhs;
    float zenith = sqrtf(cos_theta);
    float radiance_lhs = 1.0f + p[0] * expf(p[1] / (cos_theta + 0.01f));
    float radiance_rhs = p[2] + p[3] * exp_m + p[5] * ray_m + p[6] * mie_m + p[7] * zenith;
    float radiance_dist = radiance_lhs * radiance_rhs;
    return r * radiance_dist;
}
static inline double tm1(double x) {                     /* uncharted2_tonemap (:51-66), per component */
    const double a = 0.15, b = 0.50, c = 0.10, d = 0.20, e = 0.02, f = 0.30;
    double num = x * (a * x + c * b) + d * e;
    double den = x * (a * x + b) + d * f;
    return num / den - e / f;
}
static inline v3 uncharted2(v3 x) {                      /* :70-79 */
    const double exposure_bias = 1.1, w = 11.2;
    v3 curr = V(tm1(exposure_bias * x.x), tm1(exposure_bias * x.y), tm1(exposure_bias * x.z));
    double ws = 1.0 / tm1(w);
    return V(ws * curr.x, ws * curr.y, ws * curr.z);
}
void orc_uncharted2(const double x[3], double out[3]) { to3(uncharted2(from3(x)), out); }

static inline v3 sky_color_toward(const orc_scene* S, v3 dir) {   /* :84-93 */
    float theta = (float)acos(dir.z);
    double g = vdot(dir, from3(S->sky.sun));
    if (g < -1.0) g = -1.0;
    if (g > 1.0) g = 1.0;
    float gamma = (float)g;
    v3 c = V((double)orc_sky_radiance(&S->sky, theta, gamma, 0),
             (double)orc_sky_radiance(&S->sky, theta, gamma, 1),
             (double)orc_sky_radiance(&S->sky, theta, gamma, 2));
    return uncharted2(c);
}
void orc_sky_color(const orc_scene* s, const double d[3], double out[3]) { to3(sky_color_toward(s, from3(d)), out); }

/* ------------------------------------------------------------------------------------------ */
/* Camera + integrator (src/camera.rs)                                                          */
/* ------------------------------------------------------------------------------------------ */
typedef struct { const orc_scene* S; const rt_camera* cam; uint64_t seed; int mode; uint64_t segments; } tracer;

/* Camera::get_ray (:231-254); Vec3::random_in_unit_disc (src/vec3.rs:124-135), capped at 32 rejections */
static ray_t get_ray(const rt_camera* cam, uint32_t x, uint32_t y, double ox, double oy, uint64_t seed,
                     uint32_t pixel, uint32_t sample) {
    v3 p00 = from3(cam->pixel00), du = from3(cam->pixel_du), dv = from3(cam->pixel_dv), center = from3(cam->center);
    v3 ps = vadd(vadd(p00, vscale(du, (double)x + ox)), vscale(dv, (double)y + oy));
    v3 origin = center;
    if (!(cam->defocus_angle <= 0.0)) {
        v3 v = V(1.0, 1.0, 1.0);
        for (uint32_t it = 0; it < 32 && vnorm2(v) > 1.0; ++it) {
            double a = orc_uniform(seed, pixel, sample, DEPTH_PRIMARY, 2 * it);
            double b = orc_uniform(seed, pixel, sample, DEPTH_PRIMARY, 2 * it + 1);
            v = V(2.0 * a - 1.0, 2.0 * b - 1.0, 0.0);
        }
        /* defocus_disk_sample (:367-371) */
        origin = vadd(vadd(center, vscale(from3(cam->defocus_u), v.x)), vscale(from3(cam->defocus_v), v.y));
    }
    return ray_new(origin, vsub(ps, origin));
}

/* Camera::raycast (:296-313) with russian_roulette (:280-293) */
static v3 raycast(tracer* T, const ray_t* r, uint32_t depth, uint32_t pixel, uint32_t sample) {
    isect h;
    T->segments++;
    if (world_hit(T->S, r, T->cam->t_min, T->cam->t_max, T->mode, &h) > 0) {
        rng_key k = {T->seed, pixel, sample, depth};
        v3 att; ray_t sc;
        scatter(T->S, r, &h, &k, &att, &sc);
        if (depth < T->cam->max_depth) {
            double p = fmax(fmax(att.x, att.y), att.z);             /* Vector3::max() */
            if (rng_u(&k, SLOT_ROULETTE) < p) {                      /* gen_bool(p) */
                v3 w = vdivs(vscale(att, 1.0), p);                   /* ray_color * 1.0 / p */
                v3 bounced = raycast(T, &sc, depth + 1, pixel, sample);
                return vmul(w, bounced);
            }
        }
        return V(0.0, 0.0, 0.0);
    }
    v3 dir = vnormalize(r->d);
    return sky_color_toward(T->S, dir);
}

static _Atomic uint64_t g_last_segments;
uint64_t orc_last_render_segments(void) { return atomic_load(&g_last_segments); }

typedef struct {
    const orc_scene* S; const rt_camera* cam; const rt_opts* opts; int mode;
    uint32_t y0, y1; double* out; const double* h2; const double* h3;
    _Atomic uint32_t next_row; _Atomic uint64_t segments;
} render_job;

static void* render_worker(void* arg) {
    render_job* J = (render_job*)arg;
    tracer T = {J->S, J->cam, J->opts->seed, J->mode, 0};
    uint32_t W = J->cam->width;
    uint32_t stride = J->opts->sample_stride ? J->opts->sample_stride : 1;
    for (;;) {
        uint32_t y = atomic_fetch_add(&J->next_row, 1);
        if (y >= J->y1) break;
        for (uint32_t x = 0; x < W; ++x) {
            uint32_t pixel = y * W + x;
            v3 sum = V(0.0, 0.0, 0.0);                               /* render_pixel (:315-325) */
            for (uint32_t k = 0; k < J->opts->spp; ++k) {
                uint32_t s = J->opts->sample_offset + k * stride;
                ray_t r = get_ray(J->cam, x, y, J->h2[s], J->h3[s], J->opts->seed, pixel, s);
                sum = vadd(sum, raycast(&T, &r, 0, pixel, s));
            }
            v3 mean = vdivs(sum, (double)J->opts->spp);
            to3(mean, &J->out[3 * (size_t)pixel]);
        }
    }
    atomic_fetch_add(&J->segments, T.segments);
    return NULL;
}

int orc_render(const orc_scene* s, const rt_camera* cam, const rt_opts* opts, int mode, int nthreads,
               uint32_t y0, uint32_t y1, double* out, rt_stats* stats) {
    if (!s || !cam || !opts || !out || opts->spp == 0) return RT_ERR_INVALID;
    if (y1 > cam->height) y1 = cam->height;
    uint32_t stride = opts->sample_stride ? opts->sample_stride : 1;
    uint64_t last = (uint64_t)opts->sample_offset + (uint64_t)(opts->spp - 1) * stride;
    if (last >= RT_MAX_SAMPLE_INDEX) return RT_ERR_LIMIT;
    size_t len = (size_t)last + 1;
    double* h2 = (double*)malloc(len * sizeof(double));
    double* h3 = (double*)malloc(len * sizeof(double));
    if (!h2 || !h3) { free(h2); free(h3); return RT_ERR_OOM; }
    halton_fill(2, len, h2);                                        /* rng_map (:208-210) */
    halton_fill(3, len, h3);
    render_job J;
    J.S = s; J.cam = cam; J.opts = opts; J.mode = mode; J.y0 = y0; J.y1 = y1; J.out = out; J.h2 = h2; J.h3 = h3;
    atomic_init(&J.next_row, y0); atomic_init(&J.segments, 0);
    if (nthreads < 1) nthreads = 1;
    if (nthreads > 256) nthreads = 256;
    pthread_t th[256];
    int started = 0;
    for (int i = 0; i < nthreads - 1; ++i)
        if (pthread_create(&th[started], NULL, render_worker, &J) == 0) ++started;
    render_worker(&J);
    for (int i = 0; i < started; ++i) pthread_join(th[i], NULL);
    free(h2); free(h3);
    uint64_t seg = atomic_load(&J.segments);
    atomic_store(&g_last_segments, seg);
    if (stats) {
        memset(stats, 0, sizeof(*stats));
        stats->samples = (uint64_t)(y1 > y0 ? y1 - y0 : 0) * cam->width * opts->spp;
        stats->ray_segments = seg;
    }
    return RT_OK;
}

int orc_radiance_sample(const orc_scene* s, const rt_camera* cam, uint32_t x, uint32_t y, uint32_t sample_index,
                        uint64_t seed, int mode, double out_rgb[3], uint32_t* n_segments) {
    if (!s || !cam || sample_index >= RT_MAX_SAMPLE_INDEX) return RT_ERR_INVALID;
    tracer T = {s, cam, seed, mode, 0};
    uint32_t pixel = y * cam->width + x;
    ray_t r = get_ray(cam, x, y, orc_halton_closed(2, sample_index), orc_halton_closed(3, sample_index), seed, pixel, sample_index);
    to3(raycast(&T, &r, 0, pixel, sample_index), out_rgb);
    if (n_segments) *n_segments = (uint32_t)T.segments;
    return RT_OK;
}

/* ------------------------------------------------------------------------------------------ */
/* C API                                                                                        */
/* ------------------------------------------------------------------------------------------ */
static void* dup_mem(const void* p, size_t bytes) {
    void* q = malloc(bytes ? bytes : 1);
    if (q && bytes) memcpy(q, p, bytes);
    return q;
}

int orc_scene_create(const rt_sphere* spheres, size_t n_spheres, const rt_triangle* triangles, size_t n_triangles,
                     const rt_material* materials, size_t n_materials, const rt_texture* textures, size_t n_textures,
                     const uint8_t* texel_pool, size_t texel_pool_bytes, const rt_sky* sky, orc_scene** out) {
    if (!out || !sky || (n_spheres && !spheres) || (n_triangles && !triangles)) return RT_ERR_INVALID;
    orc_scene* S = (orc_scene*)calloc(1, sizeof(orc_scene));
    if (!S) return RT_ERR_OOM;
    S->nsph = n_spheres; S->ntri = n_triangles; S->nmat = n_materials; S->ntex = n_textures; S->texel_bytes = texel_pool_bytes;
    S->sph = (rt_sphere*)dup_mem(spheres, n_spheres * sizeof(rt_sphere));
    S->tri = (rt_triangle*)dup_mem(triangles, n_triangles * sizeof(rt_triangle));
    S->mat = (rt_material*)dup_mem(materials, n_materials * sizeof(rt_material));
    S->tex = (rt_texture*)dup_mem(textures, n_textures * sizeof(rt_texture));
    S->texels = (uint8_t*)dup_mem(texel_pool, texel_pool_bytes);
    S->sph_rot = (double(*)[9])malloc((n_spheres ? n_spheres : 1) * sizeof(double[9]));
    S->sky = *sky;
    if (!S->sph || !S->tri || !S->mat || !S->tex || !S->texels || !S->sph_rot) { orc_scene_destroy(S); return RT_ERR_OOM; }
    for (size_t i = 0; i < n_spheres; ++i) {
        if (S->sph[i].material >= n_materials) { orc_scene_destroy(S); return RT_ERR_INVALID; }
        sphere_rotation(S->sph[i].face, S->sph_rot[i]);
    }
    for (size_t i = 0; i < n_triangles; ++i)
        if (S->tri[i].material >= n_materials) { orc_scene_destroy(S); return RT_ERR_INVALID; }
    for (size_t i = 0; i < n_materials; ++i)
        if (S->mat[i].kind != RT_MAT_DIELECTRIC && S->mat[i].texture >= n_textures) { orc_scene_destroy(S); return RT_ERR_INVALID; }
    for (size_t i = 0; i < n_textures; ++i) {
        const rt_texture* t = &S->tex[i];
        if (t->kind == RT_TEX_CHECKER && (t->even >= i || t->odd >= i)) { orc_scene_destroy(S); return RT_ERR_INVALID; }
        if (t->kind == RT_TEX_IMAGE) {
            size_t bpt = t->texel_format == RT_TEXEL_RGBA16 ? 8 : 4;
            if (!t->width || !t->height || t->texel_offset + (size_t)t->width * t->height * bpt > texel_pool_bytes) {
                orc_scene_destroy(S); return RT_ERR_INVALID;
            }
        }
    }
    if (build_bvh(S) != 0) { orc_scene_destroy(S); return RT_ERR_OOM; }
    *out = S;
    return RT_OK;
}

void orc_scene_destroy(orc_scene* S) {
    if (!S) return;
    free(S->sph); free(S->tri); free(S->mat); free(S->tex); free(S->texels); free(S->sph_rot); free(S->nodes);
    free(S);
}

static void isect_out(const isect* h, orc_isect* o) {
    to3(h->point, o->point); to3(h->normal, o->normal);
    o->uv[0] = h->u; o->uv[1] = h->v; o->t = h->t; o->prim = h->prim; o->material = h->material; o->front = h->front; o->_pad = 0;
}

int orc_trace_closest(const orc_scene* s, const rt_ray* rays, size_t n, double t_min, double t_max, int mode,
                      rt_hit* out, orc_isect* io) {
    if (!s || (n && (!rays || !out))) return RT_ERR_INVALID;
    for (size_t i = 0; i < n; ++i) {
        ray_t r = ray_new(from3(rays[i].o), from3(rays[i].d));
        isect h;
        int f = world_hit(s, &r, t_min, t_max, mode, &h);
        if (f < 0) return RT_ERR_LIMIT;
        out[i].prim = f ? h.prim : RT_MISS;
        out[i].t = f ? h.t : 0.0;
        out[i]._pad = 0;
        if (io) { memset(&io[i], 0, sizeof(orc_isect)); if (f) isect_out(&h, &io[i]); else io[i].prim = RT_MISS; }
    }
    return RT_OK;
}

int orc_trace_primary(const orc_scene* s, const rt_camera* cam, uint32_t sample_index, int jitter, uint64_t seed,
                      int mode, rt_hit* out, rt_ray* rays_out) {
    if (!s || !cam || !out || sample_index >= RT_MAX_SAMPLE_INDEX) return RT_ERR_INVALID;
    double ox = jitter ? orc_halton_closed(2, sample_index) : 0.0;
    double oy = jitter ? orc_halton_closed(3, sample_index) : 0.0;
    rt_camera c = *cam;
    if (!jitter) c.defocus_angle = 0.0;                   /* debug_ray (:256-260) starts at the centre */
    for (uint32_t y = 0; y < cam->height; ++y)
        for (uint32_t x = 0; x < cam->width; ++x) {
            uint32_t pixel = y * cam->width + x;
            ray_t r = get_ray(&c, x, y, ox, oy, seed, pixel, sample_index);
            isect h;
            int f = world_hit(s, &r, cam->t_min, cam->t_max, mode, &h);
            if (f < 0) return RT_ERR_LIMIT;
            out[pixel].prim = f ? h.prim : RT_MISS;
            out[pixel].t = f ? h.t : 0.0;
            out[pixel]._pad = 0;
            if (rays_out) { to3(r.o, rays_out[pixel].o); to3(r.d, rays_out[pixel].d); }
        }
    return RT_OK;
}

void orc_ray_new(const double o[3], const double d[3], double out_o[3], double out_d[3]) {
    ray_t r = ray_new(from3(o), from3(d));
    to3(r.o, out_o); to3(r.d, out_d);
}
static ray_t ray_raw(const rt_ray* r) {                  /* caller passes an already-normalised ray */
    ray_t q; q.o = from3(r->o); q.d = from3(r->d); q.inv = V(1.0 / q.d.x, 1.0 / q.d.y, 1.0 / q.d.z);
    return q;
}
int orc_sphere_hit(const rt_sphere* sp, const rt_ray* ray, double lo, double hi, orc_isect* out) {
    double rot[9];
    sphere_rotation(sp->face, rot);
    ray_t r = ray_raw(ray);
    isect h;
    int f = sphere_hit(sp, rot, 0, &r, lo, hi, &h);
    if (f && out) isect_out(&h, out);
    return f;
}
int orc_triangle_hit(const rt_triangle* tr, const rt_ray* ray, double lo, double hi, orc_isect* out) {
    ray_t r = ray_raw(ray);
    isect h;
    int f = triangle_hit(tr, 0, &r, lo, hi, &h);
    if (f && out) isect_out(&h, out);
    return f;
}
int orc_scatter(const orc_scene* s, const rt_ray* ray_in, const orc_isect* hit, uint64_t seed, uint32_t pixel,
                uint32_t sample, uint32_t depth, double att[3], rt_ray* scattered) {
    if (!s || !ray_in || !hit || hit->material >= s->nmat) return RT_ERR_INVALID;
    ray_t rin = ray_raw(ray_in);
    isect h;
    h.point = from3(hit->point); h.normal = from3(hit->normal); h.t = hit->t; h.u = hit->uv[0]; h.v = hit->uv[1];
    h.prim = hit->prim; h.material = hit->material; h.front = hit->front;
    rng_key k = {seed, pixel, sample, depth};
    v3 a; ray_t sc;
    scatter(s, &rin, &h, &k, &a, &sc);
    to3(a, att); to3(sc.o, scattered->o); to3(sc.d, scattered->d);
    return RT_OK;
}

/* Vec3Ext::as_rgb_gamma (src/vec3.rs:39-42, :71-97); out-of-range clamps instead of panicking */
void orc_gamma8(const double* rgb, size_t n, uint8_t* out) {
    const double gamma = 1.0 / 2.2;
    for (size_t i = 0; i < n; ++i) {
        double c = rgb[i];
        if (!(c >= 0.0)) c = 0.0;
        if (c > 1.0) c = 1.0;
        out[i] = (uint8_t)round(pow(c, gamma) * 255.0);
    }
}
