"""Debug: lane occupancy of the bounce traversal (RT_HIP_DEBUG_UTIL) on the bench frame at low spp."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["RT_HIP_DEBUG_UTIL"] = "1"
from rt_amd import World, scenes
desc = scenes.heightfield_scene()
world = World.build(desc)
cam = scenes.heightfield_camera(1920, 1080)
for label, env in (("plain", {"RT_HIP_REFILL": "0"}), ("refill", {"RT_HIP_REFILL": "1"})):
    os.environ.update(env)
    print(label, flush=True)
    img, st = cam.render_image(world, spp=16, seed=1, with_stats=True, collect_stats=True)
    print({k: int(st[k]) for k in ("ray_segments", "primary_segments", "nodes_visited", "nodes_visited_primary")}, flush=True)
