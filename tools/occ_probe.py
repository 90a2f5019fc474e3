import os, sys
sys.path.insert(0, os.getcwd())
os.environ["RT_HIP_DEBUG_UTIL"] = "1"
from rt_amd import World, scenes
desc = scenes.heightfield_scene()
cam = scenes.heightfield_camera(1920, 1080)
w = World.build(desc)
cam.render_image(w, spp=32, seed=1)
