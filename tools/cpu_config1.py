#!/usr/bin/env python3
"""BASELINE config 1 in full on the CPU: the cover scene at 1200x675, 64 spp, depth 50, rendered by the oracle (the
reference-equivalent CPU restatement) on all host cores.  Prints one JSON line.  usage: python tools/cpu_config1.py"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import oracle_ffi as O  # noqa: E402
from rt_amd import scenes  # noqa: E402

cores = len(os.sched_getaffinity(0))
desc = scenes.cover_scene()
cam = scenes.cover_camera(1200, 675, spp=64, max_depth=50, defocus=True)
t0 = time.time()
orc = O.OracleScene(desc)
build_s = time.time() - t0
out = {}
for name, mode in (("no_pruning", O.BVH_NOPRUNE), ("distance_pruning", O.BVH_PRUNE)):
    t0 = time.time()
    img, st = orc.render(cam, 64, seed=1, mode=mode, nthreads=cores, with_stats=True)
    dt = time.time() - t0
    out[name] = {"seconds": dt, "Msamples_per_s": 1200 * 675 * 64 / dt / 1e6, "rays_per_s": float(st["ray_segments"]) / dt,
                 "image_mean": float(img.mean())}
print(json.dumps({"config": "C1 cover 1200x675 64 spp depth 50", "kind": "port (oracle/rt_oracle.c)", "cores": cores,
                  "oracle_bvh_build_s": build_s, **out}))
