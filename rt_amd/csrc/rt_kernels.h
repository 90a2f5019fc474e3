// rt_kernels.h — host-callable launchers of the wavefront kernels (rt_kernels.hip).
#pragma once
#include <hip/hip_runtime.h>

#include "rt_layout.h"

namespace rt {

// Workgroup size of the shading / streaming kernels.  One wave per workgroup: several passes are in flight and their
// kernels share the GPU; small workgroups slot in between the traversal waves (256 -> 128 -> 64 threads: 6250 -> 6550
// -> 6690 Msamples/s on the 1 M-triangle frame, and k_shade alone is 6 % faster).
#ifndef RT_BLOCK
#define RT_BLOCK 64
#endif
constexpr int BLOCK = RT_BLOCK;

// Exact u32 division by a divisor that is fixed for a pass (Granlund-Montgomery: one mul_hi, two shifts, an add and a
// subtract instead of the ~25-instruction expansion of a u32 division; the path-id decode runs once per path vertex).
struct FastDiv {
    uint32_t m, s1, s2, d;
};
FastDiv fast_div_for(uint32_t d);            // d >= 1

struct PassParams {
    rt_camera cam;
    uint64_t seed;
    uint32_t n_pixels;
    uint32_t sample_offset, sample_stride;   // global sample = offset + (pass_base + local) * stride
    uint32_t pass_base;                      // first local sample of this pass
    uint32_t jitter;                         // 0: debug_ray (pixel centres, pinhole)
    uint32_t tiled;                          // 1: path slots enumerate pixels in 8x8 tiles (coherent waves)
    uint32_t group;                          // samples of a pixel per wave of camera rays (rt_device.h: pid_to_slot_sample)
    // derived by finish_pass_params() from the fields above
    uint32_t log2_p;                         // log2(64 / group)
    FastDiv div_group;                       // by n_pixels * group (the paths of one sample group; n_pixels when group <= 1)
    FastDiv div_tpr;                         // by the full 8x8 tiles per row
    FastDiv div_width;                       // by the image width
    const double* jitter_table;              // optional: Halton (2, 3) pairs of the pass's local samples, [2 * local]
};
// call after n_pixels / group / cam / tiled are set and before the struct goes to a kernel
void finish_pass_params(PassParams& pp);

// ---- device statistics: groups of STAT_GROUP u64 counters, one group per kind of launch ----------
constexpr int STAT_GROUP = 16;
enum { STAT_GROUP_PRIMARY = 0, STAT_GROUP_BOUNCE = 1 /* first-bounce traversal */, STAT_GROUP_TAIL = 2 /* persistent kernel */,
       STAT_GROUP_DEEP = 3 /* stand-alone traversal launches from depth 2 on */, STAT_GROUPS = 4 };
enum {
    STAT_NODES = 0,        // nodes visited, per lane          (collect_stats)
    STAT_PRIMS = 1,        // primitives tested, per lane      (collect_stats)
    STAT_SEGMENTS = 2,     // closest-hit queries completed    (always, k_walk)
    STAT_ITERS = 3,        // wave loop iterations             (collect_stats)
    STAT_LEAVES = 4,       // leaf steps                       (collect_stats)
    STAT_WAVE_NODES = 5,   // node records fetched per WAVE by the coherent camera-ray kernel (collect_stats)
    STAT_WAVE_PRIMS = 6    // primitive records fetched per wave by it
};
constexpr size_t STAT_BYTES = 512;                  // >= STAT_GROUPS * STAT_GROUP * 8

// the pass's Halton table (2 doubles per local sample) for PassParams::jitter_table
void launch_jitter_table(const PassParams& pp, uint32_t n_local, double* d_out, hipStream_t s);
// K0: primary rays for path ids [0, n) -> dense queue
void launch_raygen(const PassParams& pp, const QueueGeom& qg, RayQueue q, hipStream_t s);
// World::hit input for rt_trace_closest: normalise host rays into a dense queue (Ray::new)
void launch_prepare_rays(const rt_ray* rays, const QueueGeom& qg, RayQueue q, hipStream_t s);
// K1: closest hit for every entry of the queue -> hb.  n_upper: an upper bound of the entry count known to the host
// (it only sizes the launch; the kernel reads the queue's counts from device memory).  d_counters: one statistics
// group, or NULL.
// n_estimate (0 = none): the entry count the host expects; sizes the persistent grid (any real count is still served).
// ovf (may be NULL): device rows for the part of the per-lane traversal stack that does not fit the 32 LDS rows the
// kernels keep when the tree needs more (stack_overflow_rows(S) > 0): rows_cap rows of waves * 64 entries.  Without it
// such trees are walked with the whole stack in LDS (fewer waves per CU).
struct StackOverflow {
    uint32_t* rows;
    uint32_t rows_cap;     // rows available
    uint32_t waves;        // columns available / 64: the persistent launch never has more waves than this
};
constexpr uint32_t OVF_LDS_ROWS = 32;                // stack rows the overflow-stack kernels keep in LDS
// What the walk launchers keep per scene (= per device) beyond their arguments; filled and read under the device's mutex.
struct LaunchCtx {
    bool debug = false;                  // RT_HIP_DEBUG_UTIL: say on stderr what the occupancy queries answer
    bool refill = true;                  // persistent launches with lane refill (false: one chunk per wave; tests)
    unsigned refill_slots = 0;           // tests: number of persistent waves (0 = what the device holds)
    uint32_t ovf_lds_rows = OVF_LDS_ROWS;    // tests lower it so that small scenes reach the overflow rows
    bool five = false;                   // several passes in flight: the 96-register walk (k_walk5), set per call by the renderer
    unsigned walk_slots[2][2][5] = {};   // resident waves of the scene's walk kernel by [STATS][SHADE][stack size class]; 0 = not asked yet
    unsigned walk5_slots[5] = {};        // the same for k_walk5
};
uint32_t stack_overflow_rows(const DScene& S, uint32_t lds_rows);       // rows a StackOverflow buffer needs for this scene (0: none)
void launch_walk(LaunchCtx& ctx, const DScene& S, RayQueue q, const QueueGeom& qg, uint32_t n_upper, double t_min, double t_max, HitBuf hb,
                 bool stats, unsigned long long* d_counters, hipStream_t s, uint32_t n_estimate = 0, const StackOverflow* ovf = nullptr);
// K1 + K2-4, persistent: traces AND shades the queue's paths (entries at depth depth0), bounce after bounce inside
// the kernel, until every path has ended (radiance slots written)
void launch_walk_shade(LaunchCtx& ctx, const DScene& S, const PassParams& pp, uint32_t depth0, RayQueue q, const QueueGeom& qg, uint32_t n_upper,
                       double t_min, double t_max, HitBuf hb, double* d_path_radiance, uint32_t* d_hist, bool stats,
                       unsigned long long* d_counters, hipStream_t s, uint32_t n_estimate = 0, const StackOverflow* ovf = nullptr);
// K1p: closest hit for coherent waves (camera rays; always a dense queue): wave-level packet traversal through the scalar cache
// gen != NULL: the kernel generates the camera rays of that pass itself (K0 fused) and stores them in q
// fuse != NULL (needs gen): the kernel also shades depth 0 and appends the survivors to fuse->q_out (the launch
// of k_shade for depth 0 is then skipped; q and hb receive nothing but the barycentrics of triangle hits)
struct PacketFuse {
    RayQueue q_out;
    uint32_t* d_out_counts;
    double* d_path_radiance;
};
bool packet_supported(const DScene& S);
void launch_trace_packet(const DScene& S, const PassParams* gen, RayQueue q, const QueueGeom& qg, double t_min, double t_max,
                         HitBuf hb, bool stats, unsigned long long* d_counters, hipStream_t s, const PacketFuse* fuse = nullptr);
constexpr size_t QUEUE_SLACK = 64 * NSHARD;   // physical slots beyond the path count a sharded queue may need
constexpr size_t COUNTER_BYTES = COUNT_STRIDE * 4 * NSHARD;      // one set of shard counters
// K2+K3+K4: shade (miss -> sky, hit -> scatter + Russian roulette) and compact survivors into q_out (same geometry)
// n_estimate (0 = none): the entry count the host expects; it sizes the one-wave-per-wave launch, a small persistent launch
// takes what lies beyond (correct whatever the estimate).  d_hist (may be NULL): the kernel stores the real entry count there.
void launch_shade(const DScene& S, const PassParams& pp, uint32_t depth, RayQueue q_in, const QueueGeom& qg, uint32_t n_upper,
                  uint32_t n_estimate, HitBuf hb, RayQueue q_out, uint32_t* d_out_counts, double* d_path_radiance, uint32_t* d_hist,
                  hipStream_t s, uint32_t exact_waves = 0);
constexpr int HIST_DEPTHS = 32;                // queue sizes reported per pass: entries at depth 1 .. HIST_DEPTHS-1
// K5a: rgb_sum[pixel] += radiance of the pass's samples, in sample order; clears n_zero u32 at d_zero_counts (may be NULL)
void launch_accumulate(const PassParams& pp, const double* d_path_radiance, uint32_t n_local_samples,
                       double* d_rgb_sum, uint32_t* d_zero_counts, uint32_t n_zero, hipStream_t s);
// K5b: mean = sum / total_spp
void launch_resolve(const double* d_sum, double* d_mean, size_t n_values, double total_spp, hipStream_t s);
// the f32 exchange of rt_render_multi: narrow a device's f64 sums; mean = (f64)sum32 / total_spp on the root
void launch_narrow_f32(const double* d_in, float* d_out, size_t n_values, hipStream_t s);
void launch_resolve_f32(const float* d_sum, double* d_mean, size_t n_values, double total_spp, hipStream_t s);
// output stage: 8-bit gamma-encoded frame from a resolved one (src/vec3.rs:71-97)
void launch_gamma8(const double* d_mean, size_t n_values, uint8_t* d_out, hipStream_t s);
// preview blend of one pass into the float and the 8-bit linear RGBA frame (src/window.rs:277-310)
void launch_blend_pass(const double* d_pass_sum, double* d_mean_frame, uint8_t* d_rgba8, size_t n_pixels, uint32_t n_samples,
                       uint64_t total_samples, bool quantise, hipStream_t s);
// hits (SoA, dense queue) -> rt_hit[]
void launch_pack_hits(HitBuf hb, const QueueGeom& qg, rt_hit* out, hipStream_t s);
void launch_pack_rays(RayQueue q, const QueueGeom& qg, rt_ray* out, hipStream_t s);

}  // namespace rt
