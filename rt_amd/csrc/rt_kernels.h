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

struct PassParams {
    rt_camera cam;
    uint64_t seed;
    uint32_t n_pixels;
    uint32_t sample_offset, sample_stride;   // global sample = offset + (pass_base + local) * stride
    uint32_t pass_base;                      // first local sample of this pass
    uint32_t jitter;                         // 0: debug_ray (pixel centres, pinhole)
    uint32_t tiled;                          // 1: path slots enumerate pixels in 8x8 tiles (coherent waves)
    uint32_t group;                          // samples of a pixel per wave of camera rays (rt_device.h: pid_to_slot_sample)
};

// K0: primary rays for path ids [0, n_paths) -> queue
void launch_raygen(const PassParams& pp, const QueueMap& qm, RayQueue q, hipStream_t s);
// World::hit input for rt_trace_closest: normalise host rays into a queue (Ray::new)
void launch_prepare_rays(const rt_ray* rays, const QueueMap& qm, RayQueue q, hipStream_t s);
// K1: closest hit for queue[0, n)
void launch_trace(const DScene& S, RayQueue q, const QueueMap& qm, double t_min, double t_max, HitBuf hb,
                  bool stats, unsigned long long* d_counters /* [2] */, hipStream_t s);
// K1p: the same for coherent waves (camera rays in 8x8 tiles): wave-level packet traversal through the scalar cache
// gen != NULL: the kernel generates the camera rays of that pass itself (K0 fused) and stores them in q
// fuse != NULL (needs gen): the kernel also shades depth 0 and appends the survivors to fuse->q_out (the launch
// of k_shade for depth 0 is then skipped; q and hb receive nothing but the barycentrics of triangle hits)
struct PacketFuse {
    RayQueue q_out;
    ShardBases out_bases;
    uint32_t* d_out_counts;
    double* d_path_radiance;
};
bool packet_supported(const DScene& S);
void launch_trace_packet(const DScene& S, const PassParams* gen, RayQueue q, const QueueMap& qm, double t_min, double t_max,
                         HitBuf hb, bool stats, unsigned long long* d_counters, hipStream_t s, const PacketFuse* fuse = nullptr);
// sharded-queue bookkeeping (host): a dense queue of n entries; the output bases of a shade launch
// over `in`; the map of the queue that launch produced, from its NSHARD counters (32 u32 apart)
QueueMap queue_map_single(uint32_t n);
ShardBases shard_bases_for(const QueueMap& in);
QueueMap queue_map_from_counts(const ShardBases& bases, const uint32_t* counts, uint32_t* total);
constexpr size_t QUEUE_SLACK = 64 * NSHARD;   // physical slots beyond the path count a sharded queue may need
constexpr size_t COUNTER_BYTES = 128 * NSHARD;
// K2+K3+K4: shade (miss -> sky, hit -> scatter + Russian roulette) and compact survivors into q_out
void launch_shade(const DScene& S, const PassParams& pp, uint32_t depth, RayQueue q_in, const QueueMap& qm, HitBuf hb,
                  RayQueue q_out, const ShardBases& out_bases, uint32_t* d_out_counts, double* d_path_radiance, hipStream_t s);
// K5a: rgb_sum[pixel] += radiance of the pass's samples, in sample order
void launch_accumulate(const PassParams& pp, const double* d_path_radiance, uint32_t n_local_samples,
                       double* d_rgb_sum, hipStream_t s);
// K5b: mean = sum / total_spp
void launch_resolve(const double* d_sum, double* d_mean, size_t n_values, double total_spp, hipStream_t s);
// hits (SoA) -> rt_hit[]; slot i holds path id pid[i] (identity for prepare_rays / first trace)
void launch_pack_hits(HitBuf hb, const QueueMap& qm, rt_hit* out, hipStream_t s);
void launch_pack_rays(RayQueue q, const QueueMap& qm, rt_ray* out, hipStream_t s);

}  // namespace rt
