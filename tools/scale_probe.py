"""S5-sized check on one GPU (BASELINE config C5 is 8 GPUs; this runs its scene at low spp on one): ~20 M triangles,
3840x2160.  Reports scene creation with both builders, closest-hit parity against the oracle on sampled rays, and the
render rate."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
from rt_amd import World, scenes, _abi

t0 = time.time()
desc = scenes.textured_scene(terrain_quads=1414, instances=12500, tex_size=1024)
print("scene: %d triangles, generated in %.1f s" % (len(desc.triangles), time.time() - t0), flush=True)
cam = scenes.heightfield_camera(3840, 2160, 16, 100)
rng = np.random.default_rng(7)
rays = np.concatenate([rng.uniform(-45, 45, (4000, 2)), rng.uniform(6, 12, (4000, 1)), rng.uniform(-1, 1, (4000, 2)), -rng.uniform(0.2, 1, (4000, 1))], axis=1)
ref = None
if "--oracle" in sys.argv:
    import oracle_ffi as oracle
    t0 = time.time()
    orc = oracle.OracleScene(desc)
    ref = orc.hit(rays, mode=oracle.BVH_PRUNE)
    print("oracle BVH + 4000 hits: %.1f s" % (time.time() - t0), flush=True)
for b in ("sah", "lbvh"):
    os.environ["RT_HIP_BUILDER"] = b
    t0 = time.time()
    world = World.build(desc)
    info = world.info()
    print(b, "create %.2f s (build %.2f upload %.2f) nodes %d depth %d width %d node_bytes %d device %.0f MB" % (
        time.time() - t0, info["build_seconds"], info["upload_seconds"], info["n_nodes"], info["bvh_depth"], info["bvh_width"],
        info["node_bytes"], info["device_bytes"] / 2 ** 20), flush=True)
    got = world.hit(rays)
    if ref is not None:
        bad = np.flatnonzero((got["prim"] != ref["prim"]) | (got["t"] != ref["t"]))
        print("   hits vs oracle: %d of %d differ; hit fraction %.2f" % (bad.size, len(rays), (got["prim"] != _abi.RT_MISS).mean()), flush=True)
    cam.render_image(world, spp=16, seed=1)
    t0 = time.time()
    img, st = cam.render_image(world, spp=32, seed=1, with_stats=True)
    dt = time.time() - t0
    print("   4K 32 spp: %.1f ms = %.0f Msamples/s, %.2f Grays/s, mean %.6f" % (dt * 1e3, 3840 * 2160 * 32 / dt / 1e6, int(st["ray_segments"]) / dt / 1e9, img.mean()), flush=True)
    world.close()
