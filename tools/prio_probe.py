"""Small render with the side-stream priorities on/off, with a watchdog that dumps the Python stack if it stalls."""
import faulthandler, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
faulthandler.dump_traceback_later(40, exit=True)
import numpy as np
from rt_amd import World, scenes
desc = scenes.heightfield_scene(quads=96)
world = World.build(desc)
cam = scenes.heightfield_camera(256, 144, 64, 50)
print("built", flush=True)
for i in range(3):
    t0 = time.time()
    img, st = cam.render_image(world, spp=64, seed=1, max_paths_in_flight=256 * 144 * 8, with_stats=True)
    print(i, "render ok", time.time() - t0, int(st["passes"]), int(st["ray_segments"]), img.mean(), flush=True)
