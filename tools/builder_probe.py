"""Debug: host SAH build against the device LBVH build (RT_HIP_BUILDER=lbvh): build time, tree, render rate."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rt_amd import World, scenes
desc = scenes.heightfield_scene()
cam = scenes.heightfield_camera(1920, 1080)
for b in ("sah", "lbvh"):
    os.environ["RT_HIP_BUILDER"] = b
    t0 = time.time()
    world = World.build(desc)
    t1 = time.time()
    info = world.info()
    cam.render_image(world, spp=64, seed=1)
    t2 = time.time()
    img, st = cam.render_image(world, spp=128, seed=1, with_stats=True)
    dt = time.time() - t2
    print(b, "create %.3f s (build %.3f, upload %.3f) nodes %d depth %d width %d builder %d | 128 spp %.1f ms = %.0f Msamples/s" % (
        t1 - t0, info["build_seconds"], info["upload_seconds"], info["n_nodes"], info["bvh_depth"], info["bvh_width"], info["builder"],
        dt * 1e3, 1920 * 1080 * 128 / dt / 1e6), flush=True)
