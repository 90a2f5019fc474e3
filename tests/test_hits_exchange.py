"""Parity exchange files (rt_amd/hits_exchange.py <-> integration/rust/dump_hits.rs): the formats round-trip, a
scene rebuilt from the file is bit-identical in every field the closest-hit path reads, and the comparison accepts
equal dumps and flags a single changed bit.  The oracle stands in for the Rust dumper here (no cargo in the image)."""
import numpy as np
import pytest

from rt_amd import _abi, hits_exchange as hx, scenes
from tests.oracle_ffi import OracleScene, BVH_PRUNE


def test_exchange_round_trip_and_compare(tmp_path):
    desc = scenes.mixed_test_scene()
    cam = scenes.cover_camera(48, 27, defocus=False)
    rays = hx.pixel_centre_rays(cam)
    assert rays.shape == (48 * 27, 6)
    np.testing.assert_array_equal(rays[5 * 48 + 7], cam.debug_ray(7, 5))
    rays = np.concatenate([rays, rays[:64] * np.array([1, 1, 1, 1, 1, -1.0])])      # some rays into the sky: misses
    p = tmp_path / "scene.rthx"
    hx.write_exchange(p, desc, rays, 0.001, 1.0e30)
    ex = hx.read_exchange(p)
    assert (ex["t_min"], ex["t_max"]) == (0.001, 1.0e30)
    np.testing.assert_array_equal(ex["rays"], rays)

    back = hx.desc_from_exchange(ex)                      # what the Rust side builds through the reference's constructors
    for k in ("center", "radius", "face"):
        np.testing.assert_array_equal(back.spheres[k], desc.spheres[k])
    for k in ("a", "b", "c", "uv"):                       # (this scene's stored normals predate a vertex shift: not compared)
        np.testing.assert_array_equal(back.triangles[k], desc.triangles[k])
    hf = scenes.heightfield_scene(8)                      # normals: Triangle::new_with_uv's, recomputed from the vertices
    q = tmp_path / "hf.rthx"
    hx.write_exchange(q, hf, rays[:4])
    np.testing.assert_array_equal(hx.desc_from_exchange(hx.read_exchange(q)).triangles["n"], hf.triangles["n"])

    theirs = OracleScene(back).hit(ex["rays"], ex["t_min"], ex["t_max"], mode=BVH_PRUNE)
    ours = OracleScene(desc).hit(rays, 0.001, 1.0e30)
    assert (ours["prim"] != _abi.RT_MISS).sum() > 100 and (ours["prim"] == _abi.RT_MISS).sum() > 0
    d1, d2 = tmp_path / "a.rthd", tmp_path / "b.rthd"
    hx.write_dump(d1, ours)
    hx.write_dump(d2, theirs)
    a, b = hx.read_dump(d1), hx.read_dump(d2)
    n, bad = hx.compare_dumps(a, b)
    assert n == len(rays) and len(bad) == 0
    miss = a["prim"] == _abi.RT_MISS
    assert (a["t_bits"][miss] == 0).all()

    c = b.copy()
    i = int(np.nonzero(~miss)[0][3])
    c["t_bits"][i] ^= 1                                   # one ulp
    _, bad = hx.compare_dumps(a, c)
    assert list(bad) == [i]


@pytest.mark.gpu
def test_compare_command_against_library(tmp_path, capsys):
    """`python -m rt_amd.hits_exchange compare` end to end: library hits vs a dump (here the oracle's) of the same file."""
    desc = scenes.cover_scene()
    cam = scenes.cover_camera(160, 90, defocus=False)
    x, d = tmp_path / "cover.rthx", tmp_path / "cover.rthd"
    hx.write_exchange(x, desc, hx.pixel_centre_rays(cam), 0.001, float(cam.rec["t_max"][0]))
    ex = hx.read_exchange(x)
    hx.write_dump(d, OracleScene(hx.desc_from_exchange(ex)).hit(ex["rays"], ex["t_min"], ex["t_max"], mode=BVH_PRUNE))
    assert hx._main(["hits_exchange", "compare", str(x), str(d)]) == 0
    assert f"{160 * 90} rays: {160 * 90} identical" in capsys.readouterr().out


def test_render_dump_round_trip_and_compare(tmp_path):
    """RTRD0001 (integration/rust/dump_render.rs): the header rebuilds the very same camera and sky records, the
    image round-trips, and the comparison separates bit-identical values from values beyond the tolerance."""
    cam12 = (13.0, 2.0, 3.0, 0.0, 0.5, 0.0, 0.0, 1.0, 0.0, 20.0, 10.0, 0.6)
    sky = scenes.cover_scene().sky
    img = np.random.default_rng(3).random((9, 16, 3))
    p = tmp_path / "r.rtrd"
    hx.write_render_dump(p, img, 4, 7, 99, cam12, sky)
    rd = hx.read_render_dump(p)
    assert (rd["width"], rd["height"], rd["spp"], rd["max_depth"], rd["seed"]) == (16, 9, 4, 7, 99)
    np.testing.assert_array_equal(rd["image"], img)
    assert rd["sky"].tobytes() == np.asarray(sky).tobytes()
    from rt_amd import api
    want = api.Camera.new(cam12[0:3], cam12[3:6], cam12[6:9], 10.0, 0.6, 16, 9, 4, 7, 20.0)
    assert hx.camera_from_render_dump(rd).rec.tobytes() == want.rec.tobytes()

    n, same, bad = hx.compare_renders(img, rd["image"])
    assert (n, same, len(bad)) == (16 * 9 * 3, 16 * 9 * 3, 0)
    other = img.copy()
    other[2, 3, 1] *= 1.0 + 4e-16                          # an ulp or two: inside the tolerance, not bit-identical
    other[4, 5, 2] *= 1.0 + 1e-5
    n, same, bad = hx.compare_renders(img, other)
    assert same == n - 2 and list(bad) == [(4 * 16 + 5) * 3 + 2]
    with open(p, "r+b") as f:
        f.truncate(1000)
    with pytest.raises(ValueError):
        hx.read_render_dump(p)


def test_rust_binding_source_covers_the_abi():
    """integration/rust/rt-hip-sys is source only (no cargo here): check as text that it declares every exported
    symbol, asserts the struct sizes the ctypes mirror asserts, and carries the ABI version of the header."""
    import os
    import re
    from rt_amd import _lib
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    src = open(os.path.join(root, "integration/rust/rt-hip-sys/src/lib.rs")).read()
    declared = set(re.findall(r"pub fn (rt_\w+)\(", src))
    assert declared == set(_lib.EXPORTS)
    header = open(os.path.join(root, "include/rt_hip.h")).read()
    version = re.search(r"#define RT_ABI_VERSION (\d+)u", header).group(1)
    assert f"RT_ABI_VERSION: u32 = {version};" in src
    sizes = dict(re.findall(r"size_of::<Rt(\w+)>\(\) == (\d+)", src))
    rust_name = {"rt_sphere": "Sphere", "rt_triangle": "Triangle", "rt_material": "Material", "rt_texture": "Texture",
                 "rt_sky": "Sky", "rt_camera": "Camera", "rt_opts": "Opts", "rt_stats": "Stats", "rt_ray": "Ray",
                 "rt_hit": "Hit", "rt_scene_info": "SceneInfo"}
    for c_name, (dt, size) in _abi.EXPECTED_SIZES.items():
        assert int(sizes[rust_name[c_name]]) == size == dt.itemsize, c_name
    patch = open(os.path.join(root, "integration/rust/patches/rt-counter-rng.patch")).read()
    removed = [l for l in patch.splitlines() if l.startswith("-") and "thread_rng()" in l]
    assert len(removed) == 6                               # camera.rs:288, :369; material.rs:98, :112, :164, :170


@pytest.mark.gpu
def test_compare_render_command_against_library(tmp_path, capsys):
    """`python -m rt_amd.hits_exchange compare-render` end to end; the oracle's render stands in for the patched
    reference's dump (no cargo in the image), so this pins the tool, not the reference."""
    desc = scenes.cover_scene()
    x, d = tmp_path / "cover.rthx", tmp_path / "cover.rtrd"
    cam12 = (13.0, 2.0, 3.0, 0.0, 0.0, 0.0, 0.0, 1.0, 0.0, 20.0, 10.0, 0.0)
    w, h, depth, seed = 64, 36, 6, 5
    from rt_amd import api
    for spp in (1, 3):
        cam = api.Camera.new(cam12[0:3], cam12[3:6], cam12[6:9], 10.0, 0.0, w, h, spp, depth, 20.0)
        hx.write_exchange(x, desc, hx.pixel_centre_rays(cam)[:1], 0.001, float(cam.rec["t_max"][0]))
        back = hx.desc_from_exchange(hx.read_exchange(x), desc.sky)
        img = OracleScene(back).render(cam, spp, seed=seed)
        hx.write_render_dump(d, np.asarray(img).reshape(h, w, 3), spp, depth, seed, cam12, desc.sky)
        assert hx._main(["hits_exchange", "compare-render", str(x), str(d)]) == 0
        out = capsys.readouterr().out
        assert "0 beyond rtol" in out
