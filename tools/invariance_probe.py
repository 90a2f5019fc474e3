"""Debug: the bench frame rendered under different pass sizes / passes in flight must be bit-identical."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from rt_amd import World, scenes
desc = scenes.heightfield_scene()
cam = scenes.heightfield_camera(1920, 1080)
world = World.build(desc)
spp = int(sys.argv[1]) if len(sys.argv) > 1 else 256
ref = None
for pif, pipes in ((27, 1), (27, 2), (26, 1), (26, 4), (26, 4), (25, 8), (26, 2)):
    os.environ["RT_HIP_PATHS_IN_FLIGHT"] = str(pif)
    os.environ["RT_HIP_PIPES"] = str(pipes)
    img, st = cam.render_image(world, spp=spp, seed=2024, with_stats=True)
    if ref is None:
        ref = img
    d = np.abs(img - ref)
    print("2^%d x %d pipes: mean %.16f segments %d passes %d  max|diff| %.3e  pixels differing %d" % (
        pif, pipes, img.mean(), int(st["ray_segments"]), int(st["passes"]), d.max(), int((d.max(axis=2) > 0).sum())), flush=True)
