import os, sys
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), 'tests'))
import numpy as np
from rt_amd import World, scenes
from test_gpu_parity import random_rays, degenerate_rays
desc = scenes.mixed_test_scene(seed=5, n_spheres=60, quads=40)
w32 = World.build(desc)
os.environ["RT_HIP_NODE_BOXES"] = "f64"
w64 = World.build(desc)
rr = random_rays(40000, 17)
_, a = w32.hit(rr, with_stats=True); _, b = w64.hit(rr, with_stats=True)
print("random", a["nodes_visited"], b["nodes_visited"], a["prims_tested"], b["prims_tested"])
dr = degenerate_rays(desc, 6)
k = np.arange(len(dr)) % 8
for c in range(8):
    _, a = w32.hit(dr[k == c], with_stats=True); _, b = w64.hit(dr[k == c], with_stats=True)
    print("class", c, a["nodes_visited"], b["nodes_visited"], a["prims_tested"], b["prims_tested"])
