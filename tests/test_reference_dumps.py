"""Drop-in pinning of the oracle and the library against the REAL reference.

The reference is a Rust crate and this image has no cargo, so parity with the crate itself is unpinned.
Whoever has cargo closes that gap by committing files next to this test's golden directory — nothing else:

    tests/golden/ref_<name>.rthx    the exchange file (scene + rays) written by `python -m rt_amd.hits_exchange write ...`
    tests/golden/ref_<name>.rthd    `World::hit` of the unmodified crate for those rays (integration/rust/dump_hits.rs:
                                     src/hittable.rs:137-148), as (prim, t bits)
    tests/golden/ref_<name>.rtrd    optional: a render of the crate with the counter-RNG patch applied
                                     (integration/rust/dump_render.rs + patches/rt-counter-rng.patch: src/camera.rs:296-325)

Every such set found is compared: the oracle on the CPU (`-m "not gpu"`), rt_trace_closest / rt_render on the GPU
(`-m gpu`).  With no file present the tests skip and say so.  The machinery itself is exercised below with the oracle
standing in for the crate, in a temporary directory, so that the first real dump does not meet untested code."""
import glob
import os

import numpy as np
import pytest

from rt_amd import _abi, hits_exchange as hx, scenes

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
NONE_MSG = ("no reference dump under tests/golden/ (ref_*.rthx + ref_*.rthd | ref_*.rtrd): parity with the Rust crate "
            "stays unpinned — see INTEGRATION.md §5 for how to produce them where cargo exists")


def find_sets(directory=GOLDEN):
    """[(name, exchange path, hits dump path or None, render dump path or None)] for every ref_*.rthx with at least one dump."""
    out = []
    for x in sorted(glob.glob(os.path.join(directory, "ref_*.rthx"))):
        stem = x[:-5]
        d, r = stem + ".rthd", stem + ".rtrd"
        d, r = (d if os.path.exists(d) else None), (r if os.path.exists(r) else None)
        if d or r:
            out.append((os.path.basename(stem), x, d, r))
    return out


def _as_dump(hits):
    ours = np.zeros(len(hits), hx.dump_dt)
    miss = hits["prim"] == _abi.RT_MISS
    ours["prim"] = hits["prim"]
    ours["t_bits"] = np.where(miss, np.uint64(0), np.ascontiguousarray(hits["t"]).view("<u8"))
    return ours


def check_hits(trace, exchange, dump):
    """trace(desc, rays, t_min, t_max) -> hit_dt array; equal to the dump bit for bit, ties (same t bits) excepted."""
    ex = hx.read_exchange(exchange)
    theirs = hx.read_dump(dump)
    ours = _as_dump(trace(hx.desc_from_exchange(ex), ex["rays"], ex["t_min"], ex["t_max"]))
    n, bad = hx.compare_dumps(ours, theirs)
    different = [int(i) for i in bad if ours["t_bits"][i] != theirs["t_bits"][i]]
    assert not different, (f"{os.path.basename(dump)}: {len(different)} of {n} rays differ from the reference, first "
                           f"{different[:5]}: ours {ours[different[:5]]}, reference {theirs[different[:5]]}")
    return n, len(bad)


def check_render(render, exchange, dump):
    """render(desc, camera, spp, seed) -> (H, W, 3) image; equal to the dump to the render tolerance (rtol 2e-6)."""
    ex = hx.read_exchange(exchange)
    rd = hx.read_render_dump(dump)
    ours = render(hx.desc_from_exchange(ex, rd["sky"]), hx.camera_from_render_dump(rd), rd["spp"], rd["seed"])
    n, same, bad = hx.compare_renders(ours, rd["image"])
    assert len(bad) == 0, f"{os.path.basename(dump)}: {len(bad)} of {n} values beyond rtol 2e-6, first {bad[:5]}"
    return n, same


def _oracle_trace(desc, rays, t_min, t_max):
    from tests.oracle_ffi import OracleScene, BVH_NOPRUNE
    return OracleScene(desc).hit(rays, t_min, t_max, mode=BVH_NOPRUNE)      # near-first without pruning, like bvh 0.10's iterator


def _oracle_render(desc, cam, spp, seed):
    from tests.oracle_ffi import OracleScene, BVH_NOPRUNE
    return OracleScene(desc).render(cam, spp, seed=seed, mode=BVH_NOPRUNE, nthreads=8)


def _gpu_trace(desc, rays, t_min, t_max):
    from rt_amd import World
    return World.build(desc).hit(rays, t_min, t_max)


def _gpu_render(desc, cam, spp, seed):
    from rt_amd import World
    return cam.render_image(World.build(desc), spp=spp, seed=seed)


def test_oracle_against_reference_dumps():
    sets = find_sets()
    if not sets:
        pytest.skip(NONE_MSG)
    for name, x, d, r in sets:
        if d:
            check_hits(_oracle_trace, x, d)
        if r:
            check_render(_oracle_render, x, r)


@pytest.mark.gpu
def test_library_against_reference_dumps():
    sets = find_sets()
    if not sets:
        pytest.skip(NONE_MSG)
    for name, x, d, r in sets:
        if d:
            check_hits(_gpu_trace, x, d)
        if r:
            check_render(_gpu_render, x, r)


def _stand_in_set(directory, with_render=True):
    """A dump set as the Rust programs would write it, made by the oracle (the stand-in for the crate)."""
    from rt_amd import sky as sky_mod
    desc = scenes.mixed_test_scene()
    cam12 = [7.0, -6.0, 5.0, 0.0, 0.0, 0.8, 0.0, 0.0, 1.0, 35.0, 10.0, 0.0]
    from rt_amd import api
    cam = api.Camera.new(tuple(cam12[0:3]), tuple(cam12[3:6]), tuple(cam12[6:9]), cam12[10], cam12[11], 40, 24, 3, 6, cam12[9])
    x = os.path.join(directory, "ref_standin.rthx")
    hx.write_exchange(x, desc, hx.pixel_centre_rays(cam), 0.001, float(cam.rec["t_max"][0]))
    ex = hx.read_exchange(x)
    back = hx.desc_from_exchange(ex)
    hx.write_dump(os.path.join(directory, "ref_standin.rthd"), _oracle_trace(back, ex["rays"], ex["t_min"], ex["t_max"]))
    if with_render:
        sk = sky_mod.sky_synth_v1()
        img = _oracle_render(hx.desc_from_exchange(ex, sk), cam, 3, 5)
        hx.write_render_dump(os.path.join(directory, "ref_standin.rtrd"), img, 3, 6, 5, cam12, sk)
    return x


def test_dump_machinery_with_the_oracle_standing_in(tmp_path):
    """find_sets / check_hits / check_render on a set the oracle wrote: accepted as it is, rejected after one changed bit."""
    assert find_sets(str(tmp_path)) == []
    x = _stand_in_set(str(tmp_path))
    (name, x2, d, r), = find_sets(str(tmp_path))
    assert name == "ref_standin" and x2 == x and d and r
    n, n_ties = check_hits(_oracle_trace, x, d)
    assert n == 40 * 24 and n_ties == 0
    n, same = check_render(_oracle_render, x, r)
    assert n == 40 * 24 * 3 and same == n
    dump = hx.read_dump(d)
    i = int(np.nonzero(dump["prim"] != _abi.RT_MISS)[0][7])
    raw = bytearray(open(d, "rb").read())
    raw[16 + 16 * i + 8] ^= 1                             # record i: prim u32, pad u32, t bits u64 -> lowest bit of t
    open(d, "wb").write(bytes(raw))
    with pytest.raises(AssertionError):
        check_hits(_oracle_trace, x, d)


@pytest.mark.gpu
def test_library_on_a_stand_in_dump_set(tmp_path):
    """The GPU half of the same machinery: rt_trace_closest and rt_render against the stand-in set."""
    x = _stand_in_set(str(tmp_path))
    (_, _, d, r), = find_sets(str(tmp_path))
    n, _ = check_hits(_gpu_trace, x, d)
    assert n == 40 * 24
    check_render(_gpu_render, x, r)
