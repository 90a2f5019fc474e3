"""Host mirror of the reference API (rt_amd/api.py, rt_amd/scenes.py): Camera::new, Triangle::new,
flattening, and the seeded scene generators.  CPU only."""
import math

import numpy as np
import pytest

from rt_amd import _abi, scenes
from rt_amd.api import (Camera, CheckerTexture, Dielectric, ImageTexture, Lambertian, Metal, SceneDesc, SolidColor,
                        Sphere, Triangle, generate_ground_plane, triangle_normals, write_ppm)


def test_camera_new_closed_form():
    """src/camera.rs:182-206 for an axis-aligned camera: w=(0,-1,0), u=(1,0,0), v=(0,0,1)."""
    cam = Camera.new((0.0, 0.0, 0.0), (0.0, 1.0, 0.0), (0.0, 0.0, 1.0), 1.0, 0.0, 2, 2, 1, 5, 90.0)
    h = math.tan((90.0 * (math.pi / 180.0)) / 2.0)
    r = cam.rec[0]
    assert np.array_equal(r["pixel_du"], [(2.0 * h * 1.0) * 1.0 / 2.0, 0.0, 0.0])
    assert np.allclose(r["pixel_dv"], [0.0, 0.0, -h], rtol=0, atol=1e-16)
    assert np.allclose(r["pixel00"], [-h / 2.0, 1.0, h / 2.0], rtol=0, atol=1e-15)
    assert r["t_min"] == 0.001 and r["t_max"] == 1.7976931348623157e308 and r["max_depth"] == 5
    assert np.array_equal(r["defocus_u"], [0.0, 0.0, 0.0])


@pytest.mark.parametrize("name", ["cam1", "cam2", "widecam", "topdown_cam"])
def test_shipped_cameras_are_consistent(name):
    """src/scenes.rs:17-125: the pixel grid is orthogonal, centred on the look-at ray at focus distance."""
    cam = getattr(scenes, name)(800, 600)
    r = cam.rec[0]
    du, dv, c, p00 = (np.array(r[k]) for k in ("pixel_du", "pixel_dv", "center", "pixel00"))
    assert abs(du @ dv) < 1e-15
    assert np.linalg.norm(du) == pytest.approx(np.linalg.norm(dv), rel=1e-12)       # square pixels
    centre = p00 + du * 399.5 + dv * 299.5
    lookat = {"cam1": (0, 0, 0), "cam2": (0, 0, 0), "widecam": (0, 0, 5), "topdown_cam": (0, 0, 0)}[name]
    to_look = np.array(lookat, float) - c
    assert np.allclose(np.cross(centre - c, to_look), 0.0, atol=1e-9)
    assert du[2] == pytest.approx(0.0, abs=1e-15)                                      # up = +Z: rows are horizontal
    if name in ("cam2", "topdown_cam"):
        assert r["defocus_angle"] == 0.7 and np.linalg.norm(r["defocus_u"]) > 0
    else:
        assert r["defocus_angle"] == 0.0


def test_triangle_new_normal_scalar_and_vectorised_agree():
    """src/hittable.rs:169-178."""
    a, b, c = (0.3, -1.2, 0.5), (2.0, 0.1, 0.7), (-0.4, 1.5, 3.0)
    t = Triangle(a, b, c, Lambertian.new_rgb_solid(1, 1, 1))
    n = triangle_normals(np.array([a]), np.array([b]), np.array([c]))[0]
    assert tuple(n) == t.normal
    assert np.linalg.norm(n) == pytest.approx(1.0, abs=1e-15)
    assert Triangle.new_opposite_normal(a, b, c, t.material).normal == pytest.approx(tuple(-n), abs=1e-15)
    assert (t.uv_a, t.uv_b, t.uv_c) == ((0.0, 0.0), (1.0, 0.0), (0.5, 1.0))


def test_ground_plane_winding():
    """src/scenes.rs:310-342: top_is_up => stored normals +Z."""
    m = Lambertian.new_rgb_solid(1, 1, 1)
    up = generate_ground_plane(10.0, 10.0, -0.2, m, True)
    dn = generate_ground_plane(10.0, 10.0, -0.2, m, False)
    assert all(t.normal == (0.0, 0.0, 1.0) for t in up) and all(t.normal[2] == -1.0 for t in dn)


def test_flatten_dedupes_shared_materials_and_orders_texture_children_first():
    img = ImageTexture(np.zeros((2, 3, 3), np.uint8))
    chk = CheckerTexture(0.5, SolidColor((0, 0, 0)), CheckerTexture(2.0, img, SolidColor((1, 1, 1))))
    shared = Lambertian(chk)
    shapes = [Sphere((0, 0, 0), 1.0, shared), Triangle((0, 0, 0), (1, 0, 0), (0, 1, 0), shared),
              Sphere((1, 1, 1), -3.0, Metal.new_solid((0.5, 0.5, 0.5), 0.0)), Sphere((2, 2, 2), 1.0, Dielectric.new_frosted(1.5, 0.05))]
    d = SceneDesc.from_shapes(shapes)
    assert len(d.spheres) == 3 and len(d.triangles) == 1 and len(d.materials) == 3
    assert d.spheres["material"][0] == d.triangles["material"][0]
    assert d.spheres["radius"][1] == 0.0                                   # radius.max(0.0), src/hittable.rs:275
    t = d.textures
    for i in np.flatnonzero(t["kind"] == _abi.TEX_CHECKER):
        assert t["even"][i] < i and t["odd"][i] < i
    assert t["inv_scale"][np.flatnonzero(t["kind"] == _abi.TEX_CHECKER)].tolist() == [0.5, 2.0]
    metal = d.materials[d.spheres["material"][1]]
    assert metal["kind"] == _abi.MAT_METAL and metal["has_fuzz"] == 1 and metal["fuzz"] == 0.0   # Some(0.0) != None
    glass = d.materials[d.spheres["material"][2]]
    assert glass["kind"] == _abi.MAT_DIELECTRIC and glass["ior"] == 1.5 and glass["fuzz"] == 0.05
    im = t[np.flatnonzero(t["kind"] == _abi.TEX_IMAGE)[0]]
    assert (im["width"], im["height"]) == (3, 2) and d.texel_pool.nbytes == 2 * 3 * 4


def test_scene_generators_are_seeded_and_sized():
    a, b = scenes.cover_scene(), scenes.cover_scene()
    assert a.spheres.tobytes() == b.spheres.tobytes() and a.materials.tobytes() == b.materials.tobytes()
    assert 300 < len(a.spheres) <= 487 and len(a.triangles) == 2
    kinds = a.materials["kind"]
    assert (kinds == _abi.MAT_DIELECTRIC).sum() > 5 and (kinds == _abi.MAT_METAL).sum() > 20
    assert scenes.cover_scene(seed=1).spheres.tobytes() != a.spheres.tobytes()
    hf = scenes.heightfield_scene(quads=32)
    assert len(hf.triangles) == 2 * 32 * 32 and len(hf.spheres) == 0
    assert (hf.triangles["n"][:, 2] > 0).all()                             # wound to face +Z
    assert hf.triangles["material"].max() == len(hf.materials) - 1
    assert np.abs(np.linalg.norm(hf.triangles["n"], axis=1) - 1.0).max() < 1e-15


def test_full_size_heightfield_has_the_configured_triangle_count():
    hf = scenes.heightfield_scene(quads=708)
    assert len(hf.triangles) == 1_002_528


def test_ppm_header_text(tmp_path):
    """src/camera.rs:347-350."""
    p = tmp_path / "x.ppm"
    write_ppm(str(p), np.array([[[1, 2, 3], [4, 5, 6]]], np.uint8))
    assert p.read_text() == "P3\n2 1 # width, height\n255 # max color value\n1 2 3 4 5 6\n"
