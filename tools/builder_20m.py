#!/usr/bin/env python3
"""Row f1 in numbers: the host SAH builder against the device LBVH builder (RT_HIP_BUILDER=lbvh) on the config-5 scene
(~20 M triangles, 3840x2160): scene-creation time, tree, nodes and primitives visited per ray segment, render rate, and
the two images' agreement.  Prints one JSON line.  usage: python tools/builder_20m.py [--small]  (--small: the 5 M scene)"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
from rt_amd import World, scenes  # noqa: E402


def build_logged(desc):
    """World.build with the library's stderr lines captured (it writes to fd 2 directly)."""
    import tempfile
    sys.stderr.flush()
    with tempfile.TemporaryFile(mode="w+b") as tmp:
        keep = os.dup(2)
        os.dup2(tmp.fileno(), 2)
        os.environ["RT_HIP_DEBUG_UTIL"] = "1"    # the library reports the phases of scene creation on stderr
        try:
            world = World.build(desc)
        finally:
            del os.environ["RT_HIP_DEBUG_UTIL"]
            os.dup2(keep, 2)
            os.close(keep)
        tmp.seek(0)
        return world, tmp.read().decode(errors="replace")


small = "--small" in sys.argv
t0 = time.time()
desc = scenes.textured_scene() if small else scenes.textured_scene(terrain_quads=1414, instances=12500, tex_size=1024)
gen_s = time.time() - t0
W, H, SPP = (1920, 1080, 64) if small else (3840, 2160, 32)
cam = scenes.heightfield_camera(W, H, SPP, 100)
out = {"scene": "S4 textured_5m" if small else "S5 large_20m", "triangles": len(desc.triangles), "frame": [W, H], "spp": SPP,
       "scene_generation_s": gen_s, "builders": {}}
images = {}
for b in ("sah", "lbvh"):
    os.environ["RT_HIP_BUILDER"] = b
    creates = []
    for rep in range(2):  # the second creation is the one without first-touch effects
        t0 = time.time()
        world, log = build_logged(desc)
        creates.append(time.time() - t0)
        if rep == 0:
            world.close()
    phases = [ln.replace("[rt_hip] ", "") for ln in log.splitlines() if "scene creation:" in ln or "device builder:" in ln]
    info = world.info()
    cam.render_image(world, spp=SPP // 2, seed=1)
    t0 = time.time()
    img, st = cam.render_image(world, spp=SPP, seed=1, with_stats=True)
    dt = time.time() - t0
    seg = max(int(st["ray_segments"]), 1)
    images[b] = img
    out["builders"][b] = {
        "create_s": min(creates), "create_s_first": creates[0], "build_s": info["build_seconds"], "upload_s": info["upload_seconds"],
        "n_nodes": int(info["n_nodes"]), "bvh_depth": int(info["bvh_depth"]), "bvh_width": int(info["bvh_width"]),
        "node_bytes": int(info["node_bytes"]), "device_MB": info["device_bytes"] / 2 ** 20,
        "render_ms": dt * 1e3, "Msamples_per_s": W * H * SPP / dt / 1e6, "Grays_per_s": seg / dt / 1e9,
        "phases": phases,
        "image_mean": float(img.mean())}
    world.close()
    print("[%s done]" % b, file=sys.stderr, flush=True)
a, b = images["sah"], images["lbvh"]
out["images_max_rel_diff"] = float(np.max(np.abs(a - b) / np.maximum(np.abs(a), 1e-12)))
s, l = out["builders"]["sah"], out["builders"]["lbvh"]
out["lbvh_over_sah"] = {"create_s": l["create_s"] / s["create_s"], "render_time": l["render_ms"] / s["render_ms"]}
out["pays"] = bool(l["create_s"] <= 1.0 and l["render_ms"] <= 1.15 * s["render_ms"])
print(json.dumps(out))
