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
