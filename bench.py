#!/usr/bin/env python3
"""bench.py — headline benchmark: Msamples/s (and rays/s) of the wavefront path tracer.

    python bench.py --gpus N --steps K --warmup W            (N>1: launched by torch.distributed.run)

Workload (BASELINE.json configs[2], the configuration `metric` is quoted on): synthetic procedural
height-field mesh, 1 002 528 triangles, Lambertian/metal materials under the (placeholder) sky,
1920x1080, 256 spp per GPU, max depth 50.  One "step" = one full render of that frame on every GPU:
rank r renders the global sample indices r, r+N, r+2N, ... (scene replicated in each GPU's HBM), the
per-GPU radiance-sum buffers are combined with one RCCL sum-reduce, rank 0 resolves the mean.  Work
per GPU is fixed as N grows ("weak", the default): the frame gets 256*N spp in the same time.
`--scaling strong` fixes the JOB instead (BASELINE config 4: "1080p @ 1024 spp, 1/2/4/8 GPUs"): the frame's spp are
split over the ranks, ceil((spp - r) / N) for rank r, and the JSON line says "scaling": "strong".

value = W*H*global_spp / t_step / 1e6 (Msamples/s, whole job), inputs resident in HBM.

The K timed steps are K complete frames, queued one behind the other (rt_opts.flags = RT_OPT_PIPELINED: a frame's first
passes start on the GPU while the last passes of the frame before are in their sparse late bounces) and all finished —
rt_scene_sync, then the barrier — inside the timed region; RT_BENCH_PIPELINED=0 waits for every frame before queueing
the next (2-3 % slower on the headline scene).  Defaults: 10 steps after 2 warm-up steps (0.7 s timed).
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

VALU_CEILING = 0.85      # what a pure VALU loop reads on the roofline's figure (tools/micro/valu_rate2.hip under the same counters)
HBM_PEAK_GBS = 8000.0      # MI355X_MICROARCH.md: HBM3E 8 TB/s spec peak


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", default="heightfield_1m", choices=["heightfield_1m", "cover", "textured_5m", "large_20m"])
    ap.add_argument("--spp", type=int, default=None, help="samples per pixel: per GPU (weak) or of the whole job (strong); default: the config's")
    ap.add_argument("--scaling", default="weak", choices=["weak", "strong"],
                    help="weak: --spp per GPU, the job grows with N; strong: --spp for the whole job, split over the N ranks")
    ap.add_argument("--width", type=int, default=None, help="default 1920 (3840 for large_20m)")
    ap.add_argument("--height", type=int, default=None, help="default 1080 (2160 for large_20m)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-alone", action="store_true", help="skip the extra untimed one-pass-in-flight frame (profiling runs)")
    ap.add_argument("--cpu-seconds", type=float, default=15.0, help="target wall time of the CPU-baseline sample")
    return ap.parse_args()


def make_workload(name, width, height, spp):
    from rt_amd import scenes
    big = name == "large_20m"
    width, height = width or (3840 if big else 1920), height or (2160 if big else 1080)
    if name == "heightfield_1m":
        desc = scenes.heightfield_scene(quads=708)
        cam = scenes.heightfield_camera(width, height, spp or 256, 50)
        label = "S3 height-field 1002528 tris, widecam"
    elif name == "cover":
        desc = scenes.cover_scene()
        cam = scenes.cover_camera(width, height, spp or 1024, 50, defocus=True)
        label = "S1 RTIOW cover scene (447 spheres + checker ground), cam2"
    elif name == "textured_5m":
        desc = scenes.textured_scene()
        cam = scenes.heightfield_camera(width, height, spp or 1024, 50)
        label = "S4 textured terrain + icosphere instances ~5.0M tris, widecam"
    else:
        # config 5's scene and frame on ONE GPU (the config itself is 8 GPUs at 4096 spp): a line for the record,
        # run with --no-cpu-baseline (the oracle's tree over 20 M triangles takes a minute to build)
        desc = scenes.textured_scene(terrain_quads=1414, instances=12500, tex_size=1024)
        cam = scenes.heightfield_camera(width, height, spp or 64, 100)
        label = "S5 textured terrain + icosphere instances ~20.0M tris, widecam, max_depth 100"
    return desc, cam, label


def cpu_baseline(desc, cam, target_seconds):
    """Times the oracle (kind "port": C f64 restatement, per-pixel recursion, BVH with one primitive
    per leaf traversed near-first WITHOUT distance pruning, like bvh 0.10's iterator) on the host
    cores, on a bounded sample: a band of pixel rows around the image centre at 1 spp."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_ffi as O
    cores = min(os.cpu_count() or 1, int(os.environ.get("OMP_NUM_THREADS", "16") or 16))
    t0 = time.time()
    orc = O.OracleScene(desc)
    build_s = time.time() - t0
    # size the sample from a probe: the whole frame at 1 spp, or (when even that is too long) a band
    # of rows around the image centre; then as many spp as fit the time target
    mid = cam.height // 2
    t0 = time.time()
    orc.render(cam, 1, seed=1, mode=O.BVH_NOPRUNE, nthreads=cores, rows=(mid - 8, mid + 8))
    per_row = max(time.time() - t0, 1e-4) / 16.0
    rows = int(max(16, min(cam.height, target_seconds / per_row)))
    y0 = max(0, mid - rows // 2)
    y1 = min(cam.height, y0 + rows)
    cpu_spp = 1
    if rows == cam.height:
        t0 = time.time()
        orc.render(cam, 1, seed=1, mode=O.BVH_NOPRUNE, nthreads=cores, rows=(y0, y1))
        cpu_spp = int(max(1, min(1024, round(target_seconds / max(time.time() - t0, 1e-3)))))
    t0 = time.time()
    _, st = orc.render(cam, cpu_spp, seed=1, mode=O.BVH_NOPRUNE, nthreads=cores, rows=(y0, y1), with_stats=True)
    dt = time.time() - t0
    samples = (y1 - y0) * cam.width * cpu_spp
    # the same band with distance pruning, for the record (not the reference's behaviour)
    t0 = time.time()
    orc.render(cam, cpu_spp, seed=1, mode=O.BVH_PRUNE, nthreads=cores, rows=(y0, y1))
    dt_prune = time.time() - t0
    orc.close()
    return {"value": samples / dt / 1e6, "unit": "Msamples/s", "cores": cores, "kind": "port",
            "sample": f"rows {y0}..{y1} of the same {cam.width}x{cam.height} frame at {cpu_spp} spp ({samples} samples, {dt:.1f} s)",
            "rays_per_s": float(st["ray_segments"]) / dt, "oracle_bvh_build_s": build_s,
            "value_with_distance_pruning": samples / dt_prune / 1e6}


def pmc_profile(workload):
    """What the newest committed PMC summary of this workload says about the traversal kernels
    (profiles/*_traffic_<workload>.json, written by tools/pmc_summary.py from separate rocprofv3 --pmc passes of this same
    bench command; PMC passes cannot run inside the timed bench).  Returns (walk kernel record, camera kernel record,
    file name, VALU activity of the four main kernels); records are {} when there is no file."""
    import glob
    files = sorted(glob.glob(os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles",
                                          "*_traffic_%s.json" % workload.replace("_", ""))))
    if not files:
        return {}, {}, None, None
    k = json.load(open(files[-1]))["kernels"]

    def rec(pre):
        return next((v for n, v in sorted(k.items(), key=lambda x: -x[1]["launches"]) if n.startswith(pre)), {})

    walk = rec("k_walk5<") or rec("k_walk<") or rec("k_walk8<") or rec("k_trace4<")
    for r in (walk, rec("k_trace_packet")):          # summaries written before round 3 carry one (doubled) figure
        if r and "l2_fabric_bytes_per_launch_low" not in r and "fetch_bytes_per_launch_raw" in r:
            r["l2_fabric_bytes_per_launch_low"] = r["fetch_bytes_per_launch_raw"] + r.get("write_bytes_per_launch", 0.0)
            r["l2_fabric_bytes_per_launch_high"] = 2.0 * r["fetch_bytes_per_launch_raw"] + r.get("write_bytes_per_launch", 0.0)
    valu = {"k_walk": walk.get("valu_active_per_simd_cycle"),
            "k_walk_shade": (rec("k_walk_shade<") or rec("k_walk8_shade<")).get("valu_active_per_simd_cycle"),
            "k_trace_packet": rec("k_trace_packet").get("valu_active_per_simd_cycle"),
            "k_shade": rec("k_shade").get("valu_active_per_simd_cycle")}
    return walk, rec("k_trace_packet"), "profiles/" + os.path.basename(files[-1]), valu


def visible_gpus():
    """GPUs this process would see, WITHOUT touching the HIP runtime (the launcher must stay free of a GPU context: its
    children are the ranks): the KFD topology in sysfs, narrowed by HIP_VISIBLE_DEVICES / ROCR_VISIBLE_DEVICES when set.
    None when the topology cannot be read (the ranks then find out themselves)."""
    import glob
    n = 0
    files = glob.glob("/sys/class/kfd/kfd/topology/nodes/*/properties")
    if not files:
        return None
    for f in files:
        try:
            props = dict(l.split()[:2] for l in open(f) if len(l.split()) >= 2)
        except OSError:
            return None
        if int(props.get("simd_count", "0")) > 0 and int(props.get("gfx_target_version", "0")) > 0:
            n += 1
    for var in ("HIP_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
        v = os.environ.get(var)
        if v is not None:
            n = min(n, len([x for x in v.split(",") if x.strip() != ""]))
    return n


def launch_ranks(args):
    """`bench.py --gpus N` without a launcher around it (WORLD_SIZE unset): start N fresh rank processes of this same
    script, one per GPU, BEFORE anything in this process touches the GPU (a process that has initialised HIP must not
    exec, and a parent holding a HIP context would only be in the way), relay rank 0's JSON line, and exit with the
    worst exit status.  All ranks are watched together: when one exits with an error the others — which would sit in
    the rendezvous or in a collective until it times out — are terminated.  Under `python -m torch.distributed.run`
    WORLD_SIZE is set and this is skipped."""
    import socket
    import subprocess
    import threading
    n = args.gpus
    if not STANDIN and not SHARE_GPU:
        have = visible_gpus()
        if have is not None and have < n:
            raise SystemExit(f"bench.py --gpus {n}: only {have} GPU(s) visible")
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        # ROCm's legacy IPC handles do not work in this driver stack (only dmabuf IPC does): without this RCCL's
        # intra-node transport fails in hipIpcGetMemHandle.  The harness exports it already; kept for a bare shell.
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL))
    out0 = []
    reader = threading.Thread(target=lambda: out0.append(procs[0].stdout.read()), daemon=True)      # rank 0's pipe must not fill up
    reader.start()
    rcs = [None] * n
    failed = False
    while any(rc is None for rc in rcs):
        for r, p in enumerate(procs):
            if rcs[r] is None:
                rcs[r] = p.poll()
                if rcs[r] not in (None, 0) and not failed:
                    failed = True
                    for q in procs:                       # the siblings wait for a rank that is gone
                        if q.poll() is None:
                            q.terminate()
        if any(rc is None for rc in rcs):
            time.sleep(0.05)
    reader.join(timeout=10)
    sys.stdout.write((out0[0] if out0 else b"").decode("utf-8", "replace"))
    sys.stdout.flush()
    bad = [(r, rc) for r, rc in enumerate(rcs) if rc != 0]
    if bad:
        raise SystemExit(f"bench.py: rank(s) failed: {bad}")


# RT_BENCH_STANDIN=1: launcher / partition / reduce self-test on CPU (gloo, a synthetic per-sample "renderer" defined
# below).  It measures nothing and says so in its JSON line; tests/test_bench_launcher.py drives it at world size 2.
STANDIN = os.environ.get("RT_BENCH_STANDIN") == "1"
# RT_BENCH_SHARE_GPU=1: rehearsal of the N > 1 path on a box with ONE GPU — every rank renders on device 0 with the real
# kernels, and because RCCL refuses two ranks on one device the process group is gloo with the sum-reduce staged through host
# memory (rt_amd.distributed.reduce_through_host).  It checks the partition + exchange + launcher with real renders; its
# JSON line says "rehearsal": its value is N ranks time-slicing one GPU and measures nothing.
SHARE_GPU = os.environ.get("RT_BENCH_SHARE_GPU") == "1"
# RT_BENCH_PIPELINED=0 (A/B): every step waits for its frame before the next one is queued (rt_opts.flags without
# RT_OPT_PIPELINED) — the library's behaviour for hosts that do not ask for more.
PIPELINED = os.environ.get("RT_BENCH_PIPELINED", "1") != "0"


def standin_main(args, rank, world):
    import torch
    import torch.distributed as dist
    from rt_amd.distributed import render_distributed
    if os.environ.get("RT_BENCH_STANDIN_FAIL_RANK") == str(rank):      # launcher test: a rank that dies before the rendezvous
        raise SystemExit(3)
    if world > 1:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    n_val, spp = 24, args.spp or 5

    def sample_image(s):            # a deterministic "radiance" of global sample index s
        return torch.arange(n_val, dtype=torch.float64) * 0.25 + float(s * s % 7)

    def render_sum(n_spp, offset, stride, seed, out_sum):
        for k in range(n_spp):
            out_sum += sample_image(offset + k * stride)
        return None

    out = torch.zeros(n_val, dtype=torch.float64)
    t0 = time.perf_counter()
    for _ in range(args.warmup + args.steps):
        mean, _ = render_distributed(render_sum, out, spp, seed=0, scaling=args.scaling)
    dt = time.perf_counter() - t0
    if rank == 0:
        total = spp * world if args.scaling == "weak" else spp
        want = sum(sample_image(s) for s in range(total)) / float(total)
        print(json.dumps({"metric": "launcher self-test (not a measurement)", "value": 0.0, "unit": "none", "n_gpus": world,
                          "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / max(1, args.steps) * 1e3,
                          "higher_is_better": True, "scaling": args.scaling, "vs_baseline": None, "dtype": "f64", "data": "standin-selftest",
                          "config": {"workload": "synthetic per-sample images over gloo", "partition": f"sample index mod {world}",
                                     "global_spp": total},
                          "rccl_world_size": dist.get_world_size() if dist.is_initialized() else 1,
                          "standin_max_abs_err": float((mean - want).abs().max())}), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def main():
    args = parse()
    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        return launch_ranks(args)
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if STANDIN:
        return standin_main(args, rank, world)
    # stdout carries ONE JSON line and nothing else: whatever libraries print there (RCCL writes a version banner to stdout
    # when it initialises) goes to stderr from here on, and the line is written to the saved descriptor at the end
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)
    import torch
    import torch.distributed as dist
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: there is no CPU fallback for the product path")
    if SHARE_GPU:
        local_rank = 0
    if torch.cuda.device_count() <= local_rank:
        raise SystemExit(f"rank {rank}: no GPU {local_rank} (only {torch.cuda.device_count()} visible)")
    torch.cuda.set_device(local_rank)
    # RT_BENCH_FORCE_DIST=1 initialises RCCL even for one rank (exercises the collective path on a 1-GPU box)
    use_dist = world > 1 or os.environ.get("RT_BENCH_FORCE_DIST") == "1"
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        if SHARE_GPU:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
    coll_dev = "cpu" if SHARE_GPU else "cuda"        # where the small tensors of the timing collectives live
    if use_dist and not SHARE_GPU:
        # RCCL builds its communicator — and the streams it launches on — at the first collective.  That happens here, before
        # the scene exists and outside the render path: the library places its four pipe streams on hardware queues of
        # their own by looking at the streams the process holds when the scene first renders (rt_api.hip:
        # ensure_pipe_streams).
        warm = torch.zeros(16, dtype=torch.float64, device="cuda")
        dist.all_reduce(warm)
        dist.reduce(warm, dst=0)
        torch.cuda.synchronize()

    # RT_BENCH_EXTRA_STREAMS=<n> (A/B only): n used streams in the process before the library makes its own, as a host with
    # its own copy / collective streams would have; the library's stream set-up must make the frame rate independent of n
    extra_streams = [torch.cuda.Stream() for _ in range(int(os.environ.get("RT_BENCH_EXTRA_STREAMS", "0")))]
    for es in extra_streams:
        with torch.cuda.stream(es):
            torch.zeros(16, device="cuda")
    from rt_amd import World, _abi
    from rt_amd._lib import check, lib

    desc, cam, label = make_workload(args.workload, args.width, args.height, args.spp)
    spp = cam.samples_per_pixel
    t0 = time.time()
    w = World.build(desc, device=local_rank)
    info = w.info()
    create_s = time.time() - t0
    W, H = cam.width, cam.height
    rgb_sum = torch.zeros(H * W * 3, dtype=torch.float64, device="cuda")
    rgb_mean = torch.zeros_like(rgb_sum)
    L = lib()

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    from rt_amd.distributed import hip_render_sum, hip_resolve, reduce_through_host, render_distributed

    # With more than one rank the exchange step of a frame (RCCL sum-reduce over xGMI + rank 0's resolve) is queued on a
    # side stream, so the next frame's kernels do not wait for the collective: two sum buffers alternate, an event per
    # buffer says when its exchange is over.  Everything is complete inside the timed region (barrier() synchronises the
    # device).  One rank: no exchange, the frame is resolved in place.
    side = torch.cuda.Stream() if use_dist else None
    sums = [rgb_sum, torch.zeros_like(rgb_sum)] if use_dist else [rgb_sum]
    exchanged = [None] * len(sums)
    last = {"k": 0, "mean": None}

    def step(collect=False, spp_override=None):
        """One pass of the hot path: render this rank's sample indices, RCCL sum-reduce, resolve."""
        def render_sum(n_spp, offset, stride, seed, out_sum):
            return hip_render_sum(w._h, cam, n_spp, offset, stride, seed, out_sum, local_rank, collect, pipelined=PIPELINED)

        b = last["k"] % len(sums)
        last["k"] += 1
        if exchanged[b] is not None:
            torch.cuda.current_stream().wait_event(exchanged[b])      # the frame before last has left this buffer
        mean, st = render_distributed(render_sum, sums[b], spp_override or spp, seed=2024,
                                      resolve=lambda s, total: hip_resolve(s, total, local_rank), side_stream=side,
                                      scaling=args.scaling, reduce=reduce_through_host if SHARE_GPU else None)
        if side is not None:
            exchanged[b] = torch.cuda.Event()
            exchanged[b].record(side)
        if mean is not None:
            last["mean"] = mean
        return st

    for _ in range(args.warmup):
        step()
    w.sync()                                      # (the warm-up frames' statistics stay out of the timed ones)
    barrier()
    t0 = time.perf_counter()
    stats = [st_ for st_ in (step() for _ in range(args.steps)) if st_ is not None]      # (None: a rank beyond the sample count)
    # RT_OPT_PIPELINED: a step queues its frame behind the one before and returns; every frame — and its exchange — has
    # finished when the region ends: rt_scene_sync waits for the frames (and brings the statistics the calls could not
    # have yet), the barrier for the device.
    stats.append(w.sync())
    barrier()
    dt = time.perf_counter() - t0
    t = torch.tensor([dt], dtype=torch.float64, device=coll_dev)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dt = float(t.item())
    seg = torch.tensor([float(sum(int(s["ray_segments"]) for s in stats))], dtype=torch.float64, device=coll_dev)
    if world > 1:
        dist.all_reduce(seg, op=dist.ReduceOp.SUM)

    if rank == 0:
        ms_per_step = dt / args.steps * 1e3
        from rt_amd.distributed import partition
        spp_rank0, _, _, global_spp = partition(spp, 0, world, args.scaling)
        samples_per_step = W * H * global_spp
        value = samples_per_step / (dt / args.steps) / 1e6
        rays_per_s = float(seg.item()) / dt
        # ---- roofline of the dominant kernel: k_walk on the first-bounce queue (per-lane BVH traversal, one launch per
        # pass), from HIP events the library records on the stream the kernel runs on, over the timed steps of rank 0
        S = lambda key: sum(float(s_[key]) for s_ in stats)
        bounce_s, bounce_seg, bounce_launches = S("seconds_trace_bounce"), int(S("bounce_segments")), int(S("bounce_launches"))
        prim_s, prim_seg, n_passes = S("seconds_trace_primary"), int(S("primary_segments")), int(S("passes"))
        tail_s, tail_seg = S("seconds_tail"), int(S("tail_segments"))
        shade_s = S("seconds_shade")
        # Two untimed renders on THIS rank only (no collective: the other ranks are already at the final barrier):
        # visits per segment from an instrumented low-spp render of the same frame, and the kernels' durations
        # with one pass in flight (RT_HIP_PIPES=1), i.e. without other passes sharing the GPU
        scratch = torch.zeros_like(rgb_sum)
        rgb_mean.copy_(last["mean"])                 # (the device is synchronised: the last frame's exchange is over)
        img = rgb_mean.cpu().numpy()
        if os.environ.get("RT_BENCH_DUMP_IMAGE"):        # tests: the last frame's mean image (H * W * 3 float64, numpy .npy)
            import numpy as np
            np.save(os.environ["RT_BENCH_DUMP_IMAGE"], img)
        sst = hip_render_sum(w._h, cam, min(spp, 4), 0, 1, 2024, scratch, local_rank, True)
        torch.cuda.synchronize()
        alone = None
        if not args.no_alone:
            prev_pipes = os.environ.get("RT_HIP_PIPES")
            os.environ["RT_HIP_PIPES"] = "1"
            ast = hip_render_sum(w._h, cam, spp_rank0, 0, 1, 2024, scratch, local_rank, False)
            torch.cuda.synchronize()
            if prev_pipes is None:
                del os.environ["RT_HIP_PIPES"]
            else:
                os.environ["RT_HIP_PIPES"] = prev_pipes
            alone = ast
        prim_bytes = info["tri_bytes"] if len(desc.triangles) >= len(desc.spheres) else info["sphere_bytes"]
        # algorithmic bytes per segment of the dominant kernel: ray in (o, d: 48 B) + hit out (t, bu, bv, prim: 28 B) + nodes and
        # primitives visited, counted by the instrumented kernel on the same frame (bounce launches only)
        s_bseg = max(1, int(sst["bounce_segments"]))
        v_n = int(sst["nodes_visited_bounce"]) / s_bseg
        v_p = int(sst["prims_tested_bounce"]) / s_bseg
        bytes_per_seg = 48 + 28 + v_n * info["node_bytes"] + v_p * prim_bytes
        achieved = bounce_seg * bytes_per_seg / bounce_s / 1e9 if bounce_s > 0 else 0.0
        # the camera-ray kernel, per WAVE: a wave fetches each node / primitive record once for its 64 rays (scalar loads,
        # Node32 = 64 B); per ray it writes the radiance slot of a path that ends (24 B) or the next ray of one that goes
        # on (76 B: o, d, throughput, path id)
        s_pseg = max(1, int(sst["primary_segments"]))
        s_pwaves = s_pseg / 64.0
        wv_n, wv_p = int(sst["wave_node_fetches_primary"]) / s_pwaves, int(sst["wave_prim_fetches_primary"]) / s_pwaves
        a_on = min(1.0, (int(sst["ray_segments"]) - s_pseg - int(sst["tail_segments"])) / s_pseg)      # share of camera rays that go on
        p_bytes_per_wave = wv_n * 64 + wv_p * prim_bytes + 64 * (24 * (1 - a_on) + 76 * a_on)
        p_achieved = (prim_seg / 64.0) * p_bytes_per_wave / prim_s / 1e9 if prim_s > 0 else 0.0
        pmc_walk, pmc_cam, pmc_src, valu = pmc_profile(args.workload)
        roof_alone = None
        if alone is not None:
            a_s, a_seg = float(alone["seconds_trace_bounce"]), int(alone["bounce_segments"])
            a_ach = a_seg * bytes_per_seg / a_s / 1e9 if a_s > 0 else 0.0
            roof_alone = {"achieved": a_ach, "frac": a_ach / HBM_PEAK_GBS, "avg_launch_ms": a_s / max(1, int(alone["bounce_launches"])) * 1e3,
                          "launches": int(alone["bounce_launches"]), "rays_per_s": a_seg / a_s if a_s > 0 else 0.0,
                          "camera_kernel_rays_per_s": int(alone["primary_segments"]) / max(float(alone["seconds_trace_primary"]), 1e-12),
                          "tail_kernel_rays_per_s": int(alone["tail_segments"]) / max(float(alone["seconds_tail"]), 1e-12),
                          "ms_per_frame": float(alone["seconds_total"]) * 1e3}
        k_valu = (valu or {}).get("k_walk")
        seg_per_launch = bounce_seg / max(1, bounce_launches)
        insts = pmc_walk.get("valu_instructions_per_launch")
        fabric = None
        if pmc_walk.get("l2_fabric_bytes_per_launch_low") is not None:
            fabric = [pmc_walk["l2_fabric_bytes_per_launch_low"], pmc_walk["l2_fabric_bytes_per_launch_high"]]
        cam_fabric = None
        if pmc_cam.get("l2_fabric_bytes_per_launch_low") is not None:
            cam_fabric = [pmc_cam["l2_fabric_bytes_per_launch_low"], pmc_cam["l2_fabric_bytes_per_launch_high"]]
        out = {
            "metric": "Msamples/s, 1080p 1M-tri mesh scene" if args.workload == "heightfield_1m" else f"Msamples/s, {args.workload}",
            "value": value, "unit": "Msamples/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": args.scaling, "vs_baseline": None,
            "dtype": "f64", "data": "synthetic",
            "config": {"workload": label, "width": W, "height": H, "spp_per_gpu": spp_rank0, "global_spp": global_spp,
                       "max_depth": cam.max_depth, "partition": f"sample index mod {world}" + (", ceil((spp - r) / N) per rank" if args.scaling == "strong" else ""),
                       "n_prims": info["n_prims"]},
            "rays_per_s": rays_per_s, "segments_per_sample": float(seg.item()) / (samples_per_step * args.steps),
            "launches_per_step": S("launches") / args.steps, "passes_per_step": n_passes / args.steps,
            # The bound that binds.  The dominant kernel — k_walk on the first-bounce queue — runs out of VALU issue slots, not
            # out of memory bandwidth: its scene is served by L2 + Infinity Cache (roofline_hbm_nominal below prices the
            # ALGORITHMIC bytes against the HBM peak and overshoots what any memory level delivers), and rocprofv3's
            # SQ_ACTIVE_INST_VALU says the vector ALUs are busy `achieved` of every SIMD-cycle at ~3 resident waves per SIMD
            # (a wave64 VALU instruction occupies its SIMD for 4-5 cycles, f64 and f32 alike on gfx950, and a loop of nothing
            # but independent VALU instructions, four waves per SIMD, reads 0.80-0.85 on this same figure, 0.95 for the
            # quarter-rate operations: `measured_ceiling`, profiles/r03_g_instruction_costs.txt).  achieved / frac come from
            # the committed PMC summary named in `source`
            # (separate --pmc passes of this same command; counters cannot be read inside the timed run); the launch
            # duration, segments and rays/s beside them are measured live here with HIP events on the kernel's stream.
            "roofline": {"bound": "valu_issue", "achieved": k_valu, "peak": 1.0,
                         "unit": "VALU-active wave-cycles per SIMD-cycle (SQ_ACTIVE_INST_VALU x 4 / (GRBM_GUI_ACTIVE / 8 x 1024))",
                         "frac": k_valu, "traffic": None, "source": pmc_src,
                         "measured_ceiling": VALU_CEILING, "frac_of_measured_ceiling": (k_valu / VALU_CEILING) if k_valu else None,
                         "valu_active_per_simd_cycle": valu,
                         "valu_lane_utilisation": pmc_walk.get("valu_lane_utilisation"),
                         "valu_instructions_per_launch": insts,
                         # wave-level VALU instructions per 64 ray segments: the floor a faster kernel has to lower
                         # (tools/asm_blocks.py counts ~500 per wave iteration: 60 loop head + 196 node step + 248 primitive step)
                         "valu_instructions_per_64_segments": (insts / seg_per_launch * 64.0) if insts and seg_per_launch else None,
                         "kernel": "k_walk (per-lane BVH%d traversal of the first-bounce queue, one launch per pass)" % info.get("bvh_width", 4),
                         "avg_launch_ms": bounce_s / max(1, bounce_launches) * 1e3, "launches": bounce_launches,
                         "segments": bounce_seg, "nodes_per_segment": v_n, "prims_per_segment": v_p,
                         "rays_per_s": bounce_seg / bounce_s if bounce_s > 0 else 0.0,
                         "alone": roof_alone,
                         "note": "several passes run on their own streams in the timed region, so the event-timed kernel durations include time "
                                 "shared with the other passes; 'alone' is the same kernel with one pass in flight"},
            # The memory-side figure, NOMINAL: algorithmic bytes (ray in + hit out + nodes and primitives
            # visited x record size) over the live launch duration, against the 8 TB/s HBM peak.  l2_fabric_bytes is what
            # the L2 actually requested from the fabric per launch (FETCH_SIZE + WRITE_SIZE; Infinity-Cache hits included, so
            # this is L2-miss traffic, not HBM traffic; [low, high] = every read request counted as 64 / as 128 bytes:
            # profiles/r03_a_fetch_size_calibration_gather.txt).
            "roofline_hbm_nominal": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                     "frac": achieved / HBM_PEAK_GBS, "l2_fabric_bytes": fabric, "source": pmc_src,
                                     "algorithmic_bytes_per_launch": bytes_per_seg * seg_per_launch,
                                     "bytes_per_segment": bytes_per_seg, "node_bytes": info["node_bytes"], "prim_bytes": prim_bytes,
                                     "alone": ({"achieved": roof_alone["achieved"], "frac": roof_alone["frac"]} if roof_alone else None),
                                     "note": ("BVH + primitives (%.0f MB) %s") % (info["device_bytes"] / 2 ** 20,
                                              "are resident in L2 / Infinity Cache: algorithmic bytes per second may exceed the HBM peak"
                                              if info["device_bytes"] <= 256 << 20 else
                                              "exceed the 256 MB Infinity Cache: the top of the tree is served by the caches, the rest by HBM")},
            "roofline_packet": {"bound": "hbm", "achieved": p_achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                "frac": p_achieved / HBM_PEAK_GBS, "l2_fabric_bytes": cam_fabric, "nominal": True,
                                "kernel": "k_trace_packet (camera rays: raygen + packet traversal + depth-0 shading); bytes counted per WAVE "
                                          "(a wave fetches each node / primitive once for its 64 rays)",
                                "avg_launch_ms": prim_s / max(1, n_passes) * 1e3, "launches": n_passes, "segments": prim_seg,
                                "bytes_per_wave": p_bytes_per_wave, "node_fetches_per_wave": wv_n, "prim_fetches_per_wave": wv_p,
                                "nodes_per_ray": int(sst["nodes_visited_primary"]) / s_pseg,
                                "rays_per_s": prim_seg / prim_s if prim_s > 0 else 0.0},
            "tail_kernel": {"kernel": "every bounce from depth 2 on: stand-alone k_walk + k_shade pairs while the queues are large, then k_walk_shade "
                                      "(persistent trace + shade, one launch per pass) for the rest",
                            "segments": tail_seg, "avg_launch_ms": tail_s / max(1, n_passes) * 1e3,
                            "rays_per_s": tail_seg / tail_s if tail_s > 0 else 0.0},
            "trace_fraction_of_step": (prim_s + bounce_s) / dt, "shade_fraction_of_step": shade_s / dt, "tail_fraction_of_step": tail_s / dt,
            "scene": {"bvh_nodes": info["n_nodes"], "bvh_depth": info["bvh_depth"], "device_mb": info["device_bytes"] / 2 ** 20,
                      "bvh_build_s": info["build_seconds"], "upload_s": info["upload_seconds"], "create_s": create_s},
            "frames": "pipelined (RT_OPT_PIPELINED: a frame is queued behind the one before, all finished inside the timed region)" if PIPELINED
                      else "one at a time",
            "image_mean": float(img.mean()),
            "rccl_world_size": dist.get_world_size() if dist.is_initialized() and not SHARE_GPU else 1,
        }
        if SHARE_GPU:
            out["rehearsal"] = "%d ranks share ONE GPU (gloo, sum-reduce staged through host memory): exercises the N > 1 path, measures nothing" % world
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(desc, cam, args.cpu_seconds)
        sys.stdout.flush()
        os.write(json_fd, (json.dumps(out) + "\n").encode())
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
