"""Sample-partitioned multi-GPU rendering: one process per GPU, `torch.distributed` (RCCL on ROCm).

The path shards naturally: every camera sample is independent (`render_pixel` is a plain sum,
src/camera.rs:316-324).  Rank r of N renders the global sample indices r, r+N, r+2N, ... of every
pixel with the scene replicated in its HBM; the only exchange step is one sum-reduce of the
per-pixel radiance-sum buffers, after which rank 0 divides by the total sample count.

`render_sum` is injected so that the partition + reduce logic can be exercised on CPU (gloo) with a
stand-in renderer in tests; on GPUs it is `hip_render_sum` below.
"""
import ctypes as C

import torch
import torch.distributed as dist

from . import _abi


def partition(spp, rank, world, scaling="weak"):
    """(samples of this rank, sample_offset, sample_stride, samples of the whole job).

    Rank r always renders the global sample indices r, r + N, r + 2N, ... (offset = rank, stride = world), so the
    union over the ranks is the sample set 0 .. total-1 of a single-GPU render of `total` samples.
      weak   : `spp` samples PER RANK, the job renders spp * N  (work per GPU fixed as N grows);
      strong : `spp` samples for the WHOLE JOB (BASELINE config 4: 1080p at 1024 spp on 1/2/4/8 GPUs): rank r takes
               ceil((spp - r) / N) of them — the split rt_render_multi uses (rt_api.hip) — which may be uneven and is 0
               for r >= spp."""
    if scaling == "weak":
        return spp, rank, world, spp * world
    if scaling != "strong":
        raise ValueError(f"scaling must be 'weak' or 'strong', not {scaling!r}")
    mine = (spp - rank + world - 1) // world if spp > rank else 0
    return mine, rank, world, spp


def hip_render_sum(world_handle, cam, spp, sample_offset, sample_stride, seed, out_sum, device, collect_stats=False,
                   pipelined=False):
    """Adds this rank's radiance sums into the CUDA tensor out_sum through the C ABI.  pipelined: RT_OPT_PIPELINED — the
    frame is queued behind what the current stream holds and the call returns; the stream waits for the frame, the host
    does not (rt_scene_sync / World.sync does, and returns the rest of the stats)."""
    from ._lib import check, lib
    o = cam.opts(spp=spp, sample_offset=sample_offset, sample_stride=sample_stride, seed=seed, device=device,
                 collect_stats=collect_stats, pipelined=pipelined)
    st = _abi.one(_abi.stats_dt)
    stream = torch.cuda.current_stream().cuda_stream
    check(lib().rt_render_accumulate_device(world_handle, _abi.ptr(cam.rec), _abi.ptr(o), C.c_void_p(out_sum.data_ptr()),
                                            C.c_void_p(stream), _abi.ptr(st)))
    return st[0]


def hip_resolve(out_sum, total_spp, device):
    """sum / total_spp on the device through the C ABI (rt_resolve_device, src/camera.rs:324); returns a new tensor."""
    from ._lib import check, lib
    mean = torch.empty_like(out_sum)
    stream = torch.cuda.current_stream().cuda_stream
    check(lib().rt_resolve_device(C.c_void_p(out_sum.data_ptr()), C.c_void_p(mean.data_ptr()), out_sum.numel(), int(total_spp),
                                  int(device), C.c_void_p(stream)))
    return mean


def reduce_through_host(out_sum):
    """The sum-reduce to rank 0 for a process group whose backend has no device collectives (gloo): staged through host
    memory.  Only for rehearsing the N > 1 path where RCCL cannot run — several ranks sharing ONE GPU (bench.py's
    RT_BENCH_SHARE_GPU=1; RCCL refuses two ranks on one device) — never the measured path."""
    host = out_sum.cpu()
    dist.reduce(host, dst=0, op=dist.ReduceOp.SUM)
    if dist.get_rank() == 0:
        out_sum.copy_(host)


def render_distributed(render_sum, out_sum, spp, seed=0, resolve=None, side_stream=None, scaling="weak", reduce=None):
    """Every rank calls this.  render_sum(spp, offset, stride, seed, out_sum) must ADD the rank's
    per-pixel sums into out_sum (a float64 tensor, zeroed here).  Returns (mean or None, stats):
    the mean image only on rank 0.  resolve(out_sum, total_spp) -> mean: the division on rank 0 (default: a torch
    division; GPU hosts pass hip_resolve so that the library's own resolve kernel runs).
    spp / scaling: see partition() — per rank ("weak") or for the whole job ("strong").
    reduce(out_sum): the sum-reduce to rank 0 (default: dist.reduce on out_sum where it lies — RCCL for a device tensor).

    side_stream (GPU hosts rendering frame after frame): the exchange step — the sum-reduce and rank 0's resolve — is
    queued on that stream behind the render, so the next frame's kernels need not wait for the collective.  The caller
    then owns the ordering: out_sum and the returned mean belong to side_stream until the caller has synchronised with
    it (an event recorded on it, or a device synchronise), so consecutive frames alternate between two sum buffers."""
    rank = dist.get_rank() if dist.is_initialized() else 0
    world = dist.get_world_size() if dist.is_initialized() else 1
    out_sum.zero_()
    mine, off, stride, total = partition(spp, rank, world, scaling)
    stats = render_sum(mine, off, stride, seed, out_sum) if mine > 0 else None      # (a rank beyond the sample count only joins the reduce)

    def exchange():
        if dist.is_initialized():
            if reduce is not None:
                reduce(out_sum)
            else:
                dist.reduce(out_sum, dst=0, op=dist.ReduceOp.SUM)
        if rank != 0:
            return None
        return resolve(out_sum, total) if resolve else out_sum / float(total)

    if side_stream is None:
        return exchange(), stats
    side_stream.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side_stream):
        mean = exchange()
    return mean, stats
