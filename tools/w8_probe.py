"""BVH8 against BVH4 on the bench workloads: nodes / primitives per segment (instrumented render) and frame times.
usage (GPU box): python tools/w8_probe.py heightfield_1m|textured_5m|large_20m [spp]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np                           # noqa: E402
import bench                                 # noqa: E402
from rt_amd import World                     # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "heightfield_1m"
spp = int(sys.argv[2]) if len(sys.argv) > 2 else 64
desc, cam, label = bench.make_workload(name, None, None, spp)
print(label, flush=True)
imgs = {}
for fmt in ("0", "1"):
    os.environ["RT_HIP_BVH8"] = fmt
    t0 = time.time()
    w = World.build(desc)
    info = w.info()
    print(f"BVH8={fmt}: create {time.time() - t0:.2f} s (build {info['build_seconds']:.2f}, upload {info['upload_seconds']:.2f}), width {info['bvh_width']}, "
          f"node bytes {info['node_bytes']}, device MB {info['device_bytes'] / 2**20:.0f}", flush=True)
    os.environ["RT_HIP_DEBUG_UTIL"] = "1"
    os.environ["RT_HIP_PIPES"] = "1"
    _, st = cam.render_image(w, spp=4, seed=1, with_stats=True, collect_stats=True)
    os.environ.pop("RT_HIP_DEBUG_UTIL"); os.environ.pop("RT_HIP_PIPES")
    bs = max(1, int(st["bounce_segments"]))
    print(f"   first bounce: {int(st['nodes_visited_bounce']) / bs:.2f} nodes, {int(st['prims_tested_bounce']) / bs:.2f} primitives per segment; "
          f"all: {(int(st['nodes_visited']) - int(st['nodes_visited_primary'])) / max(1, int(st['ray_segments']) - int(st['primary_segments'])):.2f} nodes per non-camera segment", flush=True)
    cam.render_image(w, spp=spp, seed=2)
    best = 1e9
    for _ in range(3):
        t0 = time.time()
        img, st = cam.render_image(w, spp=spp, seed=2, with_stats=True)
        best = min(best, time.time() - t0)
    imgs[fmt] = img
    print(f"   {spp} spp: {best * 1e3:.1f} ms, {cam.width * cam.height * spp / best / 1e6:.0f} Msamples/s, {int(st['ray_segments']) / best / 1e9:.2f} Grays/s; "
          f"first bounce {int(st['bounce_segments']) / max(float(st['seconds_trace_bounce']), 1e-9) / 1e9:.2f} Grays/s in-frame, tail {int(st['tail_segments']) / max(float(st['seconds_tail']), 1e-9) / 1e9:.2f}", flush=True)
    w.close()
print("images identical:", bool(np.array_equal(imgs["0"], imgs["1"])))
