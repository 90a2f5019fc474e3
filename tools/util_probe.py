"""Debug: lane occupancy and visit counts of the bounce traversal (RT_HIP_DEBUG_UTIL) on the bench frame at
low spp, for the node formats named on the command line (default: w4 q16)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["RT_HIP_DEBUG_UTIL"] = "1"
from rt_amd import World, scenes
desc = scenes.heightfield_scene()
cam = scenes.heightfield_camera(1920, 1080)
for fmt in (sys.argv[1:] or ["w4", "q16"]):
    os.environ["RT_HIP_NODE_BOXES"] = fmt
    world = World.build(desc)
    print(fmt, world.info(), flush=True)
    cam.render_image(world, spp=16, seed=1)
    t0 = time.time()
    cam.render_image(world, spp=64, seed=1)
    print("  64 spp: %.1f ms" % ((time.time() - t0) * 1e3), flush=True)
    img, st = cam.render_image(world, spp=16, seed=1, with_stats=True, collect_stats=True)
    seg, pseg = int(st["ray_segments"]), int(st["primary_segments"])
    print("  bounce nodes/seg %.2f prims/seg %.2f" % ((int(st["nodes_visited"]) - int(st["nodes_visited_primary"])) / (seg - pseg),
          (int(st["prims_tested"]) - int(st["prims_tested_primary"])) / (seg - pseg)), flush=True)
