/* rt_hip.h — C ABI of librt_hip.so, the MI355X (gfx950) wavefront path tracer that replaces the
 * render hot path of thabnir/rt.
 *
 * The reference (a pure-Rust crate) has no FFI of its own (no `extern`, no `unsafe`,
 * `src/lib.rs:1-8` only re-exports modules), so this header *defines* the seam.  Each entry point
 * cites the reference function a Rust host would route through it.  All structs are POD,
 * little-endian, naturally aligned; no C++ or torch types cross this boundary.
 *
 * Conventions
 *   - every function returns 0 (RT_OK) on success or a negative rt_status; no exception, abort or
 *     panic crosses the ABI; rt_last_error() gives a thread-local message for the last failure.
 *   - the caller owns every input array and every output buffer; rt_scene_create copies what it
 *     needs, the returned handle owns all device memory and is immutable and shareable across
 *     threads; calls that use the same device are serialised inside the library.
 *   - primitive ids are indices into the caller's concatenated [spheres..., triangles...] order,
 *     i.e. the position in the reference's `Vec<Shape>` (src/hittable.rs:25) when the host keeps
 *     that order; RT_MISS marks "no hit".
 *   - the library never falls back to a CPU path: without a usable HIP device every compute entry
 *     point fails with RT_ERR_NO_DEVICE.
 */
#ifndef RT_HIP_H
#define RT_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RT_ABI_VERSION 2u
#define RT_MISS 0xFFFFFFFFu
/* Camera::new builds a Halton table of 1024*1024 points (src/camera.rs:208-210); a sample index
 * beyond it would index out of bounds in the reference, so it is an API limit here. */
#define RT_MAX_SAMPLE_INDEX (1u << 20)

typedef enum rt_status {
    RT_OK = 0,
    RT_ERR_INVALID = -1,    /* bad argument (null pointer, out-of-range index, bad enum)   */
    RT_ERR_NO_DEVICE = -2,  /* no HIP device / device index out of range                   */
    RT_ERR_HIP = -3,        /* a HIP runtime call failed (message has the HIP error name)  */
    RT_ERR_OOM = -4,        /* host or device allocation failed                            */
    RT_ERR_LIMIT = -5       /* a documented limit was exceeded (sample index, no-alloc size) */
} rt_status;

/* Sphere (src/hittable.rs:260-268).  `face` is `front_direction` (default (1,0,0), :278). */
typedef struct rt_sphere {
    double center[3];
    double radius;      /* clamped max(0) by Sphere::new (:275); r == 0 never hits (NaN uv) */
    double face[3];
    uint32_t material;  /* index into the material table */
    uint32_t _pad;
} rt_sphere;

/* Triangle (src/hittable.rs:153-163).  `n` is the stored face normal computed by Triangle::new
 * (:169-178); the host mirror computes it, the kernels never recompute it. */
typedef struct rt_triangle {
    double a[3], b[3], c[3];
    double uv[3][2];    /* uv_a, uv_b, uv_c */
    double n[3];
    uint32_t material;
    uint32_t _pad;
} rt_triangle;

enum { RT_MAT_LAMBERTIAN = 0, RT_MAT_METAL = 1, RT_MAT_DIELECTRIC = 2 };
/* Material (src/material.rs:10-16): Lambertian{texture} (:63-65), Metal{texture, fuzz:Option}
 * (:79-82), Dielectric{refractive_index, fuzz:Option} (:123-128). */
typedef struct rt_material {
    uint32_t kind;
    uint32_t texture;   /* texture-node index (ignored for dielectric)            */
    uint32_t has_fuzz;  /* Option::is_some(): Some(0.0) still draws random_unit   */
    uint32_t _pad;
    double fuzz;
    double ior;         /* Dielectric::refractive_index                           */
} rt_material;

enum { RT_TEX_SOLID = 0, RT_TEX_CHECKER = 1, RT_TEX_IMAGE = 2 };
enum { RT_TEXEL_RGBA8 = 0, RT_TEXEL_RGBA16 = 1 };
/* TextureEnum (src/texture.rs:12-18).  A checker's children are arbitrary texture nodes
 * (:47-48); child indices must be smaller than the node's own index (a DAG, no cycles). */
typedef struct rt_texture {
    uint32_t kind;
    uint32_t even, odd;       /* checker children                                  */
    uint32_t width, height;   /* image size in texels (>= 1)                       */
    uint32_t texel_format;    /* RT_TEXEL_*: channel / 255.0 or channel / 65535.0   */
    uint64_t texel_offset;    /* byte offset of texel (0,0) in the texel pool,
                                 row-major, 4 channels per texel (A ignored)       */
    double rgb[3];            /* solid colour                                      */
    double inv_scale;         /* checker: 1.0 / scale (src/texture.rs:54)           */
} rt_texture;

/* hw_skymodel::rgb::SkyState::raw() (27 params + 3 radiances, f32) and World::sun_direction
 * (src/hittable.rs:27-28, :35, :38). */
typedef struct rt_sky {
    float params[27];
    float radiances[3];
    double sun[3];
} rt_sky;

/* Camera (src/camera.rs:25-51) after Camera::new (:169-227). */
typedef struct rt_camera {
    double center[3];
    double pixel00[3];
    double pixel_du[3];
    double pixel_dv[3];
    double defocus_u[3];
    double defocus_v[3];
    double defocus_angle;   /* <= 0 => pinhole (src/camera.rs:247)                   */
    double t_min;           /* 0.001 in raycast (:297)                               */
    double t_max;           /* t_range.end, f64::MAX for every shipped camera        */
    uint32_t width, height;
    uint32_t max_depth;
    uint32_t _pad;
} rt_camera;

enum { RT_PRECISION_F64 = 64 };
typedef struct rt_opts {
    uint32_t spp;            /* samples per pixel rendered by THIS call                       */
    uint32_t sample_offset;  /* global sample index of local sample k is offset + k*stride    */
    uint32_t sample_stride;  /* (multi-GPU: offset = rank, stride = world size); 0 means 1    */
    uint32_t precision;      /* RT_PRECISION_F64 (0 selects it)                               */
    uint64_t seed;           /* Philox key                             */
    int32_t device;          /* HIP device ordinal                                            */
    uint32_t max_paths_in_flight; /* 0 = library default (2^26; 16 per pixel for frames beyond 4 M pixels where memory allows; smaller for short frames: 204 B of device memory per path, for four passes in flight); paths per pass */
    uint32_t collect_stats;  /* 1: count nodes_visited / prims_tested (slower kernels)        */
    uint32_t flags;          /* RT_OPT_* bits (0 = defaults; was padding before: old callers pass 0) */
} rt_opts;
/* rt_render_multi only: exchange the per-device sums as f32 (half the xGMI bytes: 24.9 MB per device at 1080p, 99.5 MB
 * at 4K); every device still accumulates its own samples in f64, only the cross-device addition is f32. */
#define RT_OPT_REDUCE_F32 1u
/* rt_render_accumulate_device only (the other entries ignore it): queue the frame and return without waiting for it.
 * `stream` is made to wait for the frame's last kernel, so whatever the caller queues on it afterwards (a reduce, a
 * resolve, a copy) sees the complete sums; the next pipelined frame of the same shape starts on the scene's streams
 * while this one's last passes are still in their sparse late bounces (1.5 ms per frame at 1080p in which the GPU is
 * otherwise nearly idle; frames shorter than four passes keep four passes in flight across frames).  The host still blocks inside a call whenever it is four passes ahead of the device.  The
 * stats of such a call hold what was known when it returned (its launches and camera rays, and the timings of the
 * passes — possibly of the frame before — it saw finish); the bounce counters and the remaining timings arrive with
 * rt_scene_sync.  Every other entry point that uses the scene's workspaces (all but rt_scene_get_info and
 * rt_trace_closest_noalloc, which has its own scratch array) waits for the queued frames first, as does rt_scene_destroy. */
#define RT_OPT_PIPELINED 2u

typedef struct rt_stats {
    uint64_t samples;        /* camera paths traced                                     */
    uint64_t ray_segments;   /* closest-hit queries: primary + bounce + tail            */
    uint64_t nodes_visited;  /* only when collect_stats                                 */
    uint64_t prims_tested;   /* only when collect_stats                                 */
    uint32_t passes, launches;
    double seconds_trace;    /* device time (HIP events on the kernels' streams) in the camera-ray kernels and the
                                stand-alone bounce traversal launches                    */
    double seconds_shade;    /* device time in the stand-alone shade + accumulate launches */
    double seconds_total;    /* host time first launch -> everything finished           */
    uint64_t trace_launches; /* number of launches timed in seconds_trace               */
    uint64_t primary_segments;     /* camera rays (depth 0) among ray_segments                 */
    double seconds_trace_primary;  /* part of seconds_trace spent on the depth-0 launches      */
    uint64_t nodes_visited_primary; /* parts of nodes_visited / prims_tested counted by the    */
    uint64_t prims_tested_primary;  /* depth-0 (camera ray) launches; only when collect_stats  */
    /* ---- ABI v2 ---- */
    uint64_t bounce_segments;      /* segments of the first-bounce traversal launches (k_walk on the depth-1 queue, one
                                      launch per pass: the dominant kernel)                    */
    double seconds_trace_bounce;   /* their event-timed duration (part of seconds_trace)       */
    uint64_t bounce_launches;
    uint64_t tail_segments;        /* segments from depth 2 on: the stand-alone k_walk + k_shade pairs of the next few
                                      depths and the persistent trace-and-shade kernel that takes the rest */
    double seconds_tail;           /* event-timed duration of all of that                      */
    uint64_t nodes_visited_bounce, prims_tested_bounce;   /* collect_stats: the stand-alone bounce launches alone */
    uint64_t wave_node_fetches_primary;  /* collect_stats: node / primitive records the coherent camera-ray kernel  */
    uint64_t wave_prim_fetches_primary;  /* fetched once per WAVE (its memory traffic; the *_primary counts are per lane) */
} rt_stats;

typedef struct rt_ray { double o[3], d[3]; } rt_ray;   /* d need not be normalised: Ray::new does */
typedef struct rt_hit { double t; uint32_t prim; uint32_t _pad; } rt_hit;

typedef struct rt_scene rt_scene;

uint32_t rt_abi_version(void);
const char* rt_last_error(void);
int rt_device_count(int* count);

/* World::build(shapes) (src/hittable.rs:33-46): flatten to SoA, build the BVH, upload to `device`.
 * texel_pool may be NULL when no image texture exists. */
int rt_scene_create(const rt_sphere* spheres, size_t n_spheres,
                    const rt_triangle* triangles, size_t n_triangles,
                    const rt_material* materials, size_t n_materials,
                    const rt_texture* textures, size_t n_textures,
                    const uint8_t* texel_pool, size_t texel_pool_bytes,
                    const rt_sky* sky, int device, rt_scene** out);
void rt_scene_destroy(rt_scene* scene);   /* NULL is a no-op */

typedef struct rt_scene_info {
    uint64_t n_prims, n_nodes, device_bytes;
    uint32_t bvh_depth, max_leaf_size;
    uint32_t node_bytes, tri_bytes, sphere_bytes;   /* per-record sizes of the HBM layout (node: the one bounce rays traverse) */
    uint32_t bvh_width;                             /* children per node of that structure: 2 or 4 */
    double build_seconds, upload_seconds;
    uint32_t builder;                               /* 0: binned SAH on the host; 1: Morton-order build on the device (RT_HIP_BUILDER=lbvh) */
    uint32_t _pad;
} rt_scene_info;
int rt_scene_get_info(const rt_scene* scene, rt_scene_info* out);

/* World::hit(ray, t_min..t_max) (src/hittable.rs:135-149) for a batch of host rays, as used by
 * Camera::debug_raycast (src/camera.rs:262-276).  Directions are normalised like Ray::new. */
int rt_trace_closest(const rt_scene* scene, const rt_ray* rays, size_t n,
                     double t_min, double t_max, rt_hit* out, rt_stats* stats /* may be NULL */);

/* The same closest-hit query for callers that cannot allocate (code under HIP graph capture, tight serving
 * loops, harnesses that forbid hipMalloc).  NO COUNTERPART IN THE REFERENCE: it is rt_trace_closest with
 *   - d_rays / d_hits in DEVICE memory (n entries each, owned by the caller),
 *   - the ray queue and hit buffer of the batch in a fixed device array inside the library (21.3 MB on every GPU
 *     the library runs on, until the process ends) instead of a workspace allocated on demand,
 *   - at most RT_NOALLOC_MAX_RAYS rays per call (RT_ERR_LIMIT beyond that; split larger batches),
 *   - the work only QUEUED on `stream` (a hipStream_t, NULL = default): it returns without synchronising, and
 *     calls no hipMalloc / hipFree / hipHostMalloc or any other allocating routine.
 * All calls on one GPU share that GPU's array: calls whose kernels can overlap on one GPU (two streams, two
 * threads) race on it — order them on one stream or with events.  The scene must have been created on its
 * device first (rt_scene_create also looks up the array's address on that device). */
#define RT_NOALLOC_MAX_RAYS (1u << 18)
int rt_trace_closest_noalloc(const rt_scene* scene, const rt_ray* d_rays, size_t n,
                             double t_min, double t_max, rt_hit* d_hits, void* stream);

/* Camera::get_ray(x, y, i) (src/camera.rs:231-254) followed by World::hit, for every pixel of the
 * camera and one global sample index; out has width*height entries, row-major.  With
 * jitter == 0 it is Camera::debug_ray (:256-260) at integer pixel coordinates. */
int rt_trace_primary(const rt_scene* scene, const rt_camera* cam, uint32_t sample_index,
                     int jitter, uint64_t seed, rt_hit* out, rt_ray* rays_out /* may be NULL */);

/* Camera::render_image (src/camera.rs:327-341): mean radiance per pixel, linear f64 RGB,
 * out_rgb_mean[(y*width + x)*3 + c]. */
int rt_render(const rt_scene* scene, const rt_camera* cam, const rt_opts* opts,
              double* out_rgb_mean, rt_stats* stats /* may be NULL */);

/* The same render, leaving per-pixel radiance SUMS in caller-provided DEVICE memory
 * (width*height*3 doubles, added to what is there) on `stream` (a hipStream_t, NULL = default)
 * — the multi-GPU entry: ranks render disjoint sample indices, sum-reduce the buffers over RCCL,
 * then call rt_resolve_device once. (Extends the `.sum::<Vec3>()` of src/camera.rs:323.)
 * The call orders its work after what is already queued on `stream`, runs up to four passes at a
 * time on streams of the scene's own (placed on different hardware queues, whatever streams the host has made), and returns when all of it has finished (the host sleeps in
 * hipStreamSynchronize meanwhile: every launch of the frame is queued up front), so the buffer is complete for any stream on return (with RT_OPT_PIPELINED in opts->flags it returns when the frame is queued instead: see there).  Calls on one
 * scene, and on one device, are serialised inside the library: the scene owns the path workspaces. */
int rt_render_accumulate_device(const rt_scene* scene, const rt_camera* cam, const rt_opts* opts,
                                double* d_rgb_sum, void* stream, rt_stats* stats);
/* Waits until every frame queued on the scene with RT_OPT_PIPELINED has finished and puts the timings and counters
 * no call has reported yet into *stats (may be NULL).  Returns at once when nothing is queued.  No counterpart in the
 * reference, whose render_image (src/camera.rs:327-341) returns the finished image. */
int rt_scene_sync(const rt_scene* scene, rt_stats* stats);
/* d_rgb_mean[i] = d_rgb_sum[i] / total_spp (src/camera.rs:324); in place allowed. */
int rt_resolve_device(const double* d_rgb_sum, double* d_rgb_mean, size_t n_values,
                      uint64_t total_spp, int device, void* stream);

/* Single-process multi-GPU form of rt_render (BASELINE north_star: sample-index partition + RCCL sum-reduce over xGMI):
 * scenes[g] is the SAME scene created on device g's GPU (rt_scene_create once per device; n_devices distinct devices).
 * One host thread per device renders the frame's samples g, g + N, g + 2N, ... of opts->spp (opts->sample_offset /
 * sample_stride are applied first; opts->device is ignored) into a device buffer of per-pixel sums; the buffers are
 * summed into scenes[0]'s device with ncclReduce (W*H*3 values, f64, or f32 with RT_OPT_REDUCE_F32; librccl.so.1 is
 * loaded on first use), divided by opts->spp there and copied to out_rgb_mean.  The sum buffers and the stream of the
 * exchange belong to the scenes (made on first use): no allocation per call after the first.  The union of the samples is that of rt_render with the same opts, so
 * the image differs only by the order of the cross-device additions (bit-identical for n_devices == 1).
 * stats: NULL or n_devices entries, one per device.  Extends `.sum::<Vec3>()` of src/camera.rs:316-324 across GPUs. */
int rt_render_multi(rt_scene* const* scenes, uint32_t n_devices, const rt_camera* cam, const rt_opts* opts,
                    double* out_rgb_mean, rt_stats* stats);

/* One sweep of the progressive preview loop (window::render_thread, src/window.rs:256-311): every pixel gets
 * render_pixel(x, y, n_samples) — Halton points 0 .. n_samples-1 again, fresh scatter randomness through `seed` —
 * and the result is blended into the frame the scene keeps ON THE DEVICE:
 *     new * (n_samples / total_samples) + old * (1 - n_samples / total_samples)            (:292-295)
 * total_samples counts this sweep (the reference's num_samples_total[i]); total_samples == n_samples starts a new
 * progression.  With RT_PASS_QUANTISE `old` is the 8-bit linear frame / 255 and the result is re-quantised with
 * as_rgb_linear, exactly the reference's accumulation (:279-310); without it `old` is a float frame kept beside it.
 * out_rgba8 (may be NULL): W*H*4 bytes, linear RGB + 0xff, the `pixels` frame of the window.  out_rgb_mean (may be
 * NULL): W*H*3 doubles, the float frame.  Frame buffers belong to the scene: no allocation per sweep. */
#define RT_PASS_QUANTISE 1u
int rt_render_pass(const rt_scene* scene, const rt_camera* cam, uint32_t n_samples, uint64_t total_samples,
                   uint64_t seed, uint32_t flags, uint8_t* out_rgba8, double* out_rgb_mean, rt_stats* stats);

/* The output stage on the device: Vec3Ext::as_rgb_gamma (src/vec3.rs:71-97) over a resolved frame in device memory
 * (e.g. rt_resolve_device's output), queued on `stream`; clamps like rt_encode_gamma8. */
int rt_encode_gamma8_device(const double* d_rgb_mean, size_t n_values, uint8_t* d_out, int device, void* stream);

/* Vec3Ext::as_rgb_gamma (src/vec3.rs:71-97) over an image on the host: round(pow(c,1/2.2)*255);
 * values outside [0,1] are clamped instead of panicking (documented deviation). */
int rt_encode_gamma8(const double* rgb_mean, size_t n_values, uint8_t* out);

#ifdef __cplusplus
}
#endif
#endif /* RT_HIP_H */
