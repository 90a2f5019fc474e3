"""Debug: bounce-traversal rate against scene size (is the floor L2 misses on the triangle array?)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["RT_HIP_PIPES"] = "1"
from rt_amd import World, scenes
cam = scenes.heightfield_camera(1920, 1080)
for quads in (44, 88, 177, 354, 708):
    desc = scenes.heightfield_scene(quads=quads)
    world = World.build(desc)
    cam.render_image(world, spp=16, seed=1)
    img, st = cam.render_image(world, spp=64, seed=1, with_stats=True)
    img, ss = cam.render_image(world, spp=16, seed=1, with_stats=True, collect_stats=True)
    seg, pseg = int(st["ray_segments"]), int(st["primary_segments"])
    bs = float(st["seconds_trace"]) - float(st["seconds_trace_primary"])
    sseg, spseg = int(ss["ray_segments"]), int(ss["primary_segments"])
    print("quads %4d tris %8d: bounce %.2f Grays/s (%.1f ms), packet %.2f Grays/s, shade %.1f ms, nodes/seg %.1f prims/seg %.2f, packet nodes/seg %.1f" % (
        quads, 2 * quads * quads, (seg - pseg) / bs / 1e9, bs * 1e3, pseg / float(st["seconds_trace_primary"]) / 1e9, float(st["seconds_shade"]) * 1e3,
        (int(ss["nodes_visited"]) - int(ss["nodes_visited_primary"])) / (sseg - spseg),
        (int(ss["prims_tested"]) - int(ss["prims_tested_primary"])) / (sseg - spseg), int(ss["nodes_visited_primary"]) / spseg), flush=True)
