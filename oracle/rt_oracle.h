/* rt_oracle.h — CPU restatement of thabnir/rt's render hot path (plain C, f64).
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product: only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load liboracle.so, and there only as
 * the checker / the timed CPU baseline.  The product path (librt_hip.so) never links, loads or
 * calls it and has no CPU fallback.
 *
 * PARITY UNPINNED against the real reference binary: the reference is a Rust crate, no
 * cargo/rustc exists in the build image or on the GPU box, and the reference ships no tests,
 * golden vectors or seeded fixtures.  The oracle is pinned instead by
 * closed-form known-answer tests derived from the reference's source text (tests/test_oracle_kat.py)
 * and by Random123's published Philox4x32-10 vectors.  Third-party behaviour it restates from
 * memory of the crates (not under /root/reference): bvh 0.10 Ray::new, nalgebra 0.33 operation
 * order, hw-skymodel 0.1.1 radiance(), rand 0.8.5 distributions — SURVEY.md Appendix B.
 *
 * The input structs are the boundary PODs of include/rt_hip.h, so oracle and product consume the
 * very same host arrays.
 */
#ifndef RT_ORACLE_H
#define RT_ORACLE_H

#include "../include/rt_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct orc_scene orc_scene;

enum { ORC_BRUTE = 0, ORC_BVH_NOPRUNE = 1, ORC_BVH_PRUNE = 2 };

typedef struct orc_isect {
    double point[3], normal[3], uv[2], t;
    uint32_t prim, material;
    int32_t front, _pad;
} orc_isect;

int orc_scene_create(const rt_sphere* spheres, size_t n_spheres,
                     const rt_triangle* triangles, size_t n_triangles,
                     const rt_material* materials, size_t n_materials,
                     const rt_texture* textures, size_t n_textures,
                     const uint8_t* texel_pool, size_t texel_pool_bytes,
                     const rt_sky* sky, orc_scene** out);
void orc_scene_destroy(orc_scene* s);

/* World::hit for a batch of rays (directions normalised like Ray::new). */
int orc_trace_closest(const orc_scene* s, const rt_ray* rays, size_t n, double t_min, double t_max,
                      int mode, rt_hit* out, orc_isect* isect_out /* may be NULL */);
/* Camera::get_ray / debug_ray for every pixel + World::hit. */
int orc_trace_primary(const orc_scene* s, const rt_camera* cam, uint32_t sample_index, int jitter,
                      uint64_t seed, int mode, rt_hit* out, rt_ray* rays_out);
/* Camera::render_image: mean radiance per pixel, f64 RGB, row-major.  Only pixel rows
 * [y0, y1) are rendered (others left untouched) so that a bounded sample of a big image can be
 * timed.  nthreads <= 0 selects 1. */
int orc_render(const orc_scene* s, const rt_camera* cam, const rt_opts* opts, int mode, int nthreads,
               uint32_t y0, uint32_t y1, double* out_rgb_mean, rt_stats* stats);
/* One path, for debugging parity: radiance of pixel (x,y), global sample index i. */
int orc_radiance_sample(const orc_scene* s, const rt_camera* cam, uint32_t x, uint32_t y,
                        uint32_t sample_index, uint64_t seed, int mode, double out_rgb[3],
                        uint32_t* n_segments);

/* ---- unit entry points for known-answer tests ---- */
double orc_halton_recurrence(uint64_t base, uint64_t index);   /* src/camera.rs:140-165, element `index` */
double orc_halton_closed(uint32_t base, uint32_t index);       /* radical inverse of index+1         */
void orc_philox4x32_10(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]);
double orc_uniform(uint64_t seed, uint32_t pixel, uint32_t sample, uint32_t depth, uint32_t slot);
void orc_ray_new(const double o[3], const double d[3], double out_o[3], double out_d[3]);
int orc_sphere_hit(const rt_sphere* sp, const rt_ray* ray_normalised, double lo, double hi, orc_isect* out);
int orc_triangle_hit(const rt_triangle* tr, const rt_ray* ray_normalised, double lo, double hi, orc_isect* out);
void orc_reflect(const double d[3], const double n[3], double out[3]);
void orc_refract(const double d[3], const double n[3], double ratio, double out[3]);
double orc_reflectance(double cosine, double ri);
void orc_texture_value(const orc_scene* s, uint32_t tex, double u, double v, const double p[3], double out[3]);
void orc_uncharted2(const double x[3], double out[3]);
void orc_sky_color(const orc_scene* s, const double dir_unit[3], double out[3]);
float orc_sky_radiance(const rt_sky* sky, float theta, float gamma, int channel);
/* scatter for a given intersection; returns attenuation + new (normalised) ray */
int orc_scatter(const orc_scene* s, const rt_ray* ray_in_normalised, const orc_isect* hit, uint64_t seed,
                uint32_t pixel, uint32_t sample, uint32_t depth, double att[3], rt_ray* scattered);
void orc_sphere_uv(const double n[3], const double face[3], double uv[2]);
void orc_gamma8(const double* rgb, size_t n_values, uint8_t* out);
uint64_t orc_last_render_segments(void);

#ifdef __cplusplus
}
#endif
#endif
