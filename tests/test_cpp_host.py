"""C++ host mirror (rt_amd/host/rt.hpp) + example driver: builds, fails loudly without a GPU, and on
the GPU renders the same image as the Python mirror of the same scene (both sit on the C ABI)."""
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "examples", "render_small")


def _build():
    import __graft_entry__ as g
    g.build_lib()
    return g.build_example()


def test_example_builds_and_reports_missing_gpu(tmp_path):
    import ctypes as C
    from rt_amd import _lib
    exe = _build()
    n = C.c_int(0)
    if _lib.lib().rt_device_count(C.byref(n)) == 0 and n.value > 0:
        pytest.skip("a GPU is present")
    r = subprocess.run([exe, str(tmp_path / "a.ppm"), str(tmp_path / "a.f64")], capture_output=True, text=True)
    assert r.returncode == 1 and "no CPU fallback" in r.stderr


@pytest.mark.gpu
def test_cpp_mirror_matches_python_mirror(tmp_path, oracle):
    from rt_amd import World
    from rt_amd.api import (Camera, CheckerTexture, Dielectric, Lambertian, Metal, SceneDesc, SolidColor, Sphere,
                            generate_ground_plane)
    assert os.path.exists(EXE), "examples/render_small must be prebuilt by __graft_entry__.build()"
    r = subprocess.run([EXE, str(tmp_path / "a.ppm"), str(tmp_path / "a.f64")], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr
    got = np.fromfile(str(tmp_path / "a.f64"), dtype=np.float64).reshape(54, 96, 3)
    cam = Camera.new((14.0, 3.0, 10.0), (0.0, 0.0, 0.0), (0.0, 0.0, 1.0), 16.0, 0.7, 96, 54, 4, 50, 20.0)
    checker = Lambertian(CheckerTexture(3.0, SolidColor((0.1, 0.1, 0.1)), SolidColor((0.95, 0.95, 0.95))))
    shapes = generate_ground_plane(10000.0, 10000.0, -0.2, checker, True)
    shapes += [Sphere((-1.0, -1.732, 0.5), 0.7, Dielectric(1.5)), Sphere((-2.0, 0.0, 0.5), 0.7, Metal.new_solid((0.7, 0.6, 0.5), None)),
               Sphere((1.0, 1.0, 0.0), 0.2, Metal.new_solid((0.2, 0.8, 0.3), 0.25)),
               Sphere.new_facing((-1.0, 1.732, 0.5), 0.7, Lambertian(SolidColor((0.4, 0.2, 0.9))), (14.0, 3.0, 10.0))]
    # the C++ flattening lists spheres first, then triangles, like SceneDesc.from_shapes
    desc = SceneDesc.from_shapes(shapes)
    img = cam.render_image(World.build(desc), spp=4, seed=7)
    assert np.array_equal(got, img)
    ref = oracle.OracleScene(desc).render(cam, 4, seed=7, mode=oracle.BRUTE)
    np.testing.assert_allclose(got, ref, rtol=2e-6, atol=1e-12)
    ppm = open(str(tmp_path / "a.ppm")).read().split("\n")
    assert ppm[0] == "P3" and ppm[1] == "96 54 # width, height" and ppm[2] == "255 # max color value"
