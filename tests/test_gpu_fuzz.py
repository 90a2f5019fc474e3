"""Seeded sweep over random small scenes and cameras: the HIP path (through the C ABI) against the CPU oracle.  Every case
draws the scene (sphere count, mesh resolution: all material and texture kinds of `mixed_test_scene`), the camera (position,
field of view, defocus disc or pinhole, frame sizes that are not multiples of the 8x8 camera tile), samples per pixel,
depth limit and RNG seed from one integer, so a failure names the case that reproduces it.  Bar as in test_gpu_parity.py:
closest hits bit-exact, equal segment counts, radiance within 2e-6 (f32 sky transcendentals)."""
import os

import numpy as np
import pytest

from rt_amd import World, scenes
from rt_amd.api import Camera

pytestmark = pytest.mark.gpu

RTOL, ATOL = 2e-6, 1e-12


def draw_case(case, large=False):
    g = np.random.default_rng((5000 if large else 1000) + case)
    n_spheres = int(g.integers(300, 2001)) if large else int(g.integers(0, 61))
    quads = int(g.integers(24, 65)) if large else int(g.integers(2, 17))
    ang, rad = g.uniform(0, 2 * np.pi), g.uniform(5.0, 14.0)
    look_from = (float(rad * np.cos(ang)), float(rad * np.sin(ang)), float(g.uniform(0.4, 8.0)))
    look_at = (float(g.uniform(-2, 2)), float(g.uniform(-2, 2)), float(g.uniform(0.0, 1.5)))
    focus = float(g.uniform(4.0, 15.0))
    defocus = float(g.choice([0.0, 0.0, 0.2, 1.0]))
    width, height = int(g.integers(33, 101)), int(g.integers(17, 71))
    spp = int(g.integers(1, 3)) if large else int(g.integers(1, 6))
    max_depth = int(g.choice([1, 2, 7, 50]))
    vfov = float(g.uniform(15.0, 70.0))
    seed = int(g.integers(0, 2**31))
    return dict(scene_seed=case + 1, n_spheres=n_spheres, quads=quads, look_from=look_from, look_at=look_at, focus=focus,
                defocus=defocus, width=width, height=height, spp=spp, max_depth=max_depth, vfov=vfov, seed=seed)


def random_rays(n, seed):
    g = np.random.default_rng(seed)
    o = g.uniform(-8.0, 8.0, (n, 3))
    o[:, 2] = g.uniform(-0.5, 6.0, n)
    d = g.uniform(-1.0, 1.0, (n, 3))
    d[np.all(d == 0.0, axis=1)] = (0.0, 0.0, -1.0)
    return np.concatenate([o, d], -1)


# RT_FUZZ_CASES=<n> widens the sweep (400 cases were run once on the final round-2 build: all equal)
@pytest.mark.parametrize("case", range(int(os.environ.get("RT_FUZZ_CASES", "24"))))
def test_random_scene_and_camera_match_oracle(oracle, case):
    c = draw_case(case)
    desc = scenes.mixed_test_scene(seed=c["scene_seed"], n_spheres=c["n_spheres"], quads=c["quads"])
    world, orc = World.build(desc), oracle.OracleScene(desc)
    try:
        rays = random_rays(4000, c["seed"] ^ 0x5bd1)
        gpu, ref = world.hit(rays), orc.hit(rays, mode=oracle.BRUTE)
        bad = np.flatnonzero((gpu["prim"] != ref["prim"]) | (gpu["t"] != ref["t"]))
        assert bad.size == 0, f"case {c}: {bad.size} closest hits differ, first gpu={gpu[bad[:2]]} ref={ref[bad[:2]]}"
        cam = Camera.new(c["look_from"], c["look_at"], (0.0, 0.0, 1.0), c["focus"], c["defocus"], c["width"], c["height"],
                         c["spp"], c["max_depth"], c["vfov"])
        img, st = cam.render_image(world, spp=c["spp"], seed=c["seed"], with_stats=True)
        want, rst = orc.render(cam, c["spp"], seed=c["seed"], mode=oracle.BRUTE, with_stats=True)
        assert int(st["samples"]) == c["width"] * c["height"] * c["spp"], c
        assert int(st["ray_segments"]) == int(rst["ray_segments"]), c
        np.testing.assert_allclose(img, want, rtol=RTOL, atol=ATOL, err_msg=str(c))
    finally:
        orc.close()


@pytest.mark.parametrize("case", range(6))
def test_random_larger_scenes_match_oracle(oracle, case):
    """The same sweep over scenes of a few thousand primitives (300-2000 overlapping spheres, meshes of up to 8192
    triangles): trees deep enough for every stack and leaf path of the walk; the oracle renders through its own BVH."""
    c = draw_case(case, large=True)
    desc = scenes.mixed_test_scene(seed=c["scene_seed"], n_spheres=c["n_spheres"], quads=c["quads"])
    world, orc = World.build(desc), oracle.OracleScene(desc)
    try:
        rays = random_rays(3000, c["seed"] ^ 0x77)
        gpu, ref = world.hit(rays), orc.hit(rays, mode=oracle.BRUTE)
        bad = np.flatnonzero((gpu["prim"] != ref["prim"]) | (gpu["t"] != ref["t"]))
        assert bad.size == 0, f"case {c}: {bad.size} closest hits differ, first gpu={gpu[bad[:2]]} ref={ref[bad[:2]]}"
        cam = Camera.new(c["look_from"], c["look_at"], (0.0, 0.0, 1.0), c["focus"], c["defocus"], c["width"], c["height"],
                         c["spp"], c["max_depth"], c["vfov"])
        img, st = cam.render_image(world, spp=c["spp"], seed=c["seed"], with_stats=True)
        want, rst = orc.render(cam, c["spp"], seed=c["seed"], mode=oracle.BVH_PRUNE, with_stats=True)
        assert int(st["ray_segments"]) == int(rst["ray_segments"]), c
        np.testing.assert_allclose(img, want, rtol=RTOL, atol=ATOL, err_msg=str(c))
    finally:
        orc.close()


@pytest.mark.parametrize("case", range(6))
def test_random_textured_meshes_match_oracle(oracle, case):
    """The sweep over the image-textured mesh path (BASELINE config 4 in small: terrain + instanced icospheres, Metal with
    image textures and fuzz on triangles, frosted dielectrics): random mesh sizes, texture sizes and cameras."""
    g = np.random.default_rng(9000 + case)
    desc = scenes.textured_scene(terrain_quads=int(g.integers(8, 41)), instances=int(g.integers(5, 121)), subdiv=int(g.integers(0, 3)),
                                 n_textures=int(g.integers(1, 6)), tex_size=int(g.choice([8, 32, 100])), seed=int(g.integers(1, 1000)))
    world, orc = World.build(desc), oracle.OracleScene(desc)
    try:
        n = 3000
        o = g.uniform(-45.0, 45.0, (n, 3))
        o[:, 2] = g.uniform(0.0, 15.0, n)
        d = g.uniform(-1.0, 1.0, (n, 3))
        d[:, 2] -= 0.5
        rays = np.concatenate([o, d], -1)
        gpu, ref = world.hit(rays), orc.hit(rays, mode=oracle.BVH_PRUNE)
        bad = np.flatnonzero((gpu["prim"] != ref["prim"]) | (gpu["t"] != ref["t"]))
        assert bad.size == 0, f"case {case}: {bad.size} closest hits differ"
        ang = g.uniform(0, 2 * np.pi)
        cam = Camera.new((float(60 * np.cos(ang)), float(60 * np.sin(ang)), float(g.uniform(8.0, 40.0))), (0.0, 0.0, 2.0), (0.0, 0.0, 1.0),
                         10.0, 0.0, int(g.integers(40, 97)), int(g.integers(24, 65)), 2, int(g.choice([3, 12, 50])), float(g.uniform(25.0, 60.0)))
        seed = int(g.integers(0, 2**31))
        img, st = cam.render_image(world, spp=2, seed=seed, with_stats=True)
        want, rst = orc.render(cam, 2, seed=seed, mode=oracle.BVH_PRUNE, with_stats=True)
        assert int(st["ray_segments"]) == int(rst["ray_segments"]), case
        np.testing.assert_allclose(img, want, rtol=RTOL, atol=ATOL)
    finally:
        orc.close()
