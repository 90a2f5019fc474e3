import faulthandler, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
faulthandler.dump_traceback_later(60, exit=True)
import numpy as np
from rt_amd import World, scenes
world = World.build(scenes.heightfield_scene(quads=96))
cam = scenes.heightfield_camera(256, 144, 64, 50)
a, sa = cam.render_image(world, spp=64, seed=1, max_paths_in_flight=256 * 144 * 8, with_stats=True)
os.environ["RT_HIP_HOST_LOOP"] = "1"
b, sb = cam.render_image(world, spp=64, seed=1, max_paths_in_flight=256 * 144 * 8, with_stats=True)
print("equal", np.array_equal(a, b), int(sa["ray_segments"]), int(sb["ray_segments"]), int(sa["launches"]), int(sb["launches"]))
