"""GPU parity tests proper: the HIP path (through the C ABI) against the CPU oracle on the same
seeded inputs.  Bar: (prim, t) bit-exact and identical segment counts (same paths: same RNG, same
discrete decisions); radiance within RTOL = 2e-6 relative: the sky model evaluates cosf/expf/powf in
f32 like hw-skymodel does (src/hittable.rs:85-90), and glibc's and the GPU's f32 transcendentals
differ in the last ulp (6e-8), which the formula amplifies a few times; everything else is f64."""
import numpy as np
import pytest

from rt_amd import World, _abi, scenes
from rt_amd.api import Camera, Lambertian, SceneDesc, Sphere, Triangle

pytestmark = pytest.mark.gpu

F64_MAX = 1.7976931348623157e308
RTOL, ATOL = 2e-6, 1e-12


def random_rays(n, seed, lo=-8.0, hi=8.0):
    k = np.arange(n)
    o = np.stack([lo + (hi - lo) * scenes.hash_u01(seed, k, c) for c in range(3)], -1)
    o[:, 2] = 0.2 + 6.0 * scenes.hash_u01(seed, k, 7)
    d = np.stack([2.0 * scenes.hash_u01(seed + 1, k, c) - 1.0 for c in range(3)], -1)
    return np.concatenate([o, d], -1)


def assert_hits_equal(gpu, ref):
    bad = np.flatnonzero((gpu["prim"] != ref["prim"]) | (gpu["t"] != ref["t"]))
    assert bad.size == 0, f"{bad.size} of {len(ref)} hits differ, first: gpu={gpu[bad[:3]]} ref={ref[bad[:3]]}"


@pytest.fixture(scope="module")
def cover(oracle):
    desc = scenes.cover_scene()
    return desc, World.build(desc), oracle.OracleScene(desc)


@pytest.fixture(scope="module")
def mixed(oracle):
    desc = scenes.mixed_test_scene()
    return desc, World.build(desc), oracle.OracleScene(desc)


def test_closest_hit_random_rays_cover(cover, oracle):
    desc, world, orc = cover
    rays = random_rays(50_000, 5)
    assert_hits_equal(world.hit(rays), orc.hit(rays, mode=oracle.BRUTE))


def test_closest_hit_random_rays_mixed(mixed, oracle):
    desc, world, orc = mixed
    rays = random_rays(30_000, 9)
    gpu = world.hit(rays)
    assert_hits_equal(gpu, orc.hit(rays, mode=oracle.BRUTE))
    assert (gpu["prim"] != _abi.RT_MISS).mean() > 0.3


def test_closest_hit_range_limits(mixed, oracle):
    desc, world, orc = mixed
    rays = random_rays(5_000, 21)
    for t_min, t_max in ((0.001, 3.0), (0.5, 7.0), (0.0, F64_MAX)):
        assert_hits_equal(world.hit(rays, t_min, t_max), orc.hit(rays, t_min, t_max, mode=oracle.BRUTE))


def test_primary_rays_debug_and_jittered(cover, oracle):
    desc, world, orc = cover
    cam = scenes.cover_camera(320, 180, defocus=False)
    g, gr = cam.trace_primary(world, 0, jitter=False, with_rays=True)
    r, rr = orc.trace_primary(cam, 0, jitter=False, mode=oracle.BRUTE, with_rays=True)
    assert np.array_equal(gr["o"], rr["o"]) and np.array_equal(gr["d"], rr["d"])
    assert_hits_equal(g, r)
    for s in (0, 1, 17, 4095):
        assert_hits_equal(cam.trace_primary(world, s, jitter=True), orc.trace_primary(cam, s, jitter=True, mode=oracle.BRUTE))


def test_primary_rays_defocus_uses_counter_rng(cover, oracle):
    desc, world, orc = cover
    cam = scenes.cover_camera(160, 90, defocus=True)
    g, gr = cam.trace_primary(world, 3, jitter=True, seed=99, with_rays=True)
    r, rr = orc.trace_primary(cam, 3, jitter=True, seed=99, mode=oracle.BRUTE, with_rays=True)
    assert np.array_equal(gr["o"], rr["o"]) and np.array_equal(gr["d"], rr["d"])
    assert_hits_equal(g, r)


def test_heightfield_bvh_vs_oracle(oracle):
    desc = scenes.heightfield_scene(quads=96)
    world, orc = World.build(desc), oracle.OracleScene(desc)
    cam = scenes.heightfield_camera(256, 144)
    assert_hits_equal(cam.trace_primary(world, 5, jitter=True), orc.trace_primary(cam, 5, jitter=True, mode=oracle.BVH_PRUNE))
    rays = random_rays(20_000, 33, -45.0, 45.0)
    assert_hits_equal(world.hit(rays), orc.hit(rays, mode=oracle.BVH_PRUNE))
    info = world.info()
    assert info["n_prims"] == 2 * 96 * 96 and info["bvh_depth"] <= 32 and info["max_leaf_size"] <= 4


@pytest.mark.parametrize("spp,seed", [(1, 0), (3, 12345)])
def test_render_matches_oracle_cover(cover, oracle, spp, seed):
    desc, world, orc = cover
    cam = scenes.cover_camera(128, 72, spp=spp, max_depth=50, defocus=True)
    img, st = cam.render_image(world, spp=spp, seed=seed, with_stats=True)
    ref, rst = orc.render(cam, spp, seed=seed, mode=oracle.BRUTE, with_stats=True)
    assert int(st["ray_segments"]) == int(rst["ray_segments"])
    assert int(st["samples"]) == 128 * 72 * spp
    np.testing.assert_allclose(img, ref, rtol=RTOL, atol=ATOL)


def test_render_matches_oracle_mixed_all_materials(mixed, oracle):
    desc, world, orc = mixed
    cam = Camera.new((9.0, -7.0, 4.0), (0.0, 0.0, 0.8), (0.0, 0.0, 1.0), 10.0, 0.3, 96, 64, 4, 12, 35.0)
    img, st = cam.render_image(world, spp=4, seed=3, with_stats=True)
    ref, rst = orc.render(cam, 4, seed=3, mode=oracle.BRUTE, with_stats=True)
    assert int(st["ray_segments"]) == int(rst["ray_segments"])
    np.testing.assert_allclose(img, ref, rtol=RTOL, atol=ATOL)


def test_render_pass_splitting_and_sample_partition_are_invariant(cover):
    """Same global sample set => same image, however it is split into passes or across 'ranks'."""
    desc, world, _ = cover
    cam = scenes.cover_camera(64, 36, spp=8, max_depth=20, defocus=True)
    full = cam.render_image(world, spp=8, seed=5)
    small_passes = cam.render_image(world, spp=8, seed=5, max_paths_in_flight=64 * 36 * 3)
    assert np.array_equal(full, small_passes)
    parts = [cam.render_image(world, spp=4, seed=5, sample_offset=r, sample_stride=2) for r in range(2)]
    np.testing.assert_allclose((parts[0] + parts[1]) / 2.0, full, rtol=1e-12, atol=1e-15)


def test_empty_and_tiny_scenes(oracle):
    cam = Camera.new((0.0, -4.0, 1.0), (0.0, 0.0, 0.0), (0.0, 0.0, 1.0), 4.0, 0.0, 33, 17, 2, 5, 40.0)
    empty = SceneDesc()
    w = World.build(empty)
    hits = cam.trace_primary(w, 0)
    assert (hits["prim"] == _abi.RT_MISS).all()
    np.testing.assert_allclose(cam.render_image(w, spp=2), oracle.OracleScene(empty).render(cam, 2, mode=oracle.BRUTE), rtol=RTOL, atol=ATOL)
    mat = Lambertian.new_rgb_solid(0.5, 0.6, 0.7)
    for shapes in ([Sphere((0, 0, 0), 1.0, mat)], [Triangle((-1, 0, -1), (1, 0, -1), (0, 0, 1), mat)],
                   [Sphere((0, 0, 0), 0.5, mat), Triangle((-2, 1, -1), (2, 1, -1), (0, 1, 2), mat)],
                   [Sphere((0, 0, 0), 0.0, mat)]):
        d = SceneDesc.from_shapes(shapes)
        w, o = World.build(d), oracle.OracleScene(d)
        assert_hits_equal(cam.trace_primary(w, 1), o.trace_primary(cam, 1, mode=oracle.BRUTE))
        np.testing.assert_allclose(cam.render_image(w, spp=2, seed=1), o.render(cam, 2, seed=1, mode=oracle.BRUTE), rtol=RTOL, atol=ATOL)


def test_stats_counters(cover):
    desc, world, _ = cover
    rays = random_rays(10_000, 2)
    _, st = world.hit(rays, with_stats=True)
    assert int(st["ray_segments"]) == 10_000 and int(st["nodes_visited"]) > 10_000 and int(st["prims_tested"]) > 0
    assert float(st["seconds_trace"]) > 0.0


def test_gpu_reproduces_golden_vectors(cover, mixed):
    """tests/golden/golden_v1.npz (made by tests/golden/make_golden.py with the oracle, brute force)."""
    import os
    from golden.make_golden import golden_inputs
    gold = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "golden_v1.npz"))
    gi = golden_inputs()
    h = gi["cover_cam"].trace_primary(cover[1], 0, jitter=True)
    assert np.array_equal(h["prim"], gold["cover_jitter_prim"]) and np.array_equal(h["t"], gold["cover_jitter_t"])
    h = gi["cover_cam"].trace_primary(cover[1], 0, jitter=False)
    assert np.array_equal(h["prim"], gold["cover_debug_prim"]) and np.array_equal(h["t"], gold["cover_debug_t"])
    h = mixed[1].hit(gi["mixed_rays"])
    assert np.array_equal(h["prim"], gold["mixed_prim"]) and np.array_equal(h["t"], gold["mixed_t"])
    hf = World.build(scenes.heightfield_scene(quads=48))
    h = gi["hf_cam"].trace_primary(hf, 3, jitter=True)
    assert np.array_equal(h["prim"], gold["hf_prim"]) and np.array_equal(h["t"], gold["hf_t"])
    img, st = gi["mixed_cam"].render_image(mixed[1], spp=2, seed=3, with_stats=True)
    assert int(st["ray_segments"]) == int(gold["mixed_render_segments"][0])
    np.testing.assert_allclose(img, gold["mixed_render"], rtol=RTOL, atol=ATOL)


def test_full_size_1m_triangle_frame_properties(oracle):
    """BASELINE.json configs[2] at full size (1 002 528 triangles, 1920x1080): the oracle checks a
    strided subset of the primary rays bit-exactly; the whole frame is checked through
    size-independent properties (hit points lie on their triangle's plane and inside it,
    determinism, sample-partition invariance, value range)."""
    desc = scenes.heightfield_scene(quads=708)
    world = World.build(desc)
    cam = scenes.heightfield_camera(1920, 1080, 2, 50)
    hits, rays = cam.trace_primary(world, 11, jitter=True, with_rays=True)
    sub = np.arange(0, len(rays), 37)[:60_000]
    orc = oracle.OracleScene(desc)
    # (both sides re-normalise the already unit-length directions, like Ray::new would)
    assert_hits_equal(world.hit(rays[sub]), orc.hit(rays[sub], mode=oracle.BVH_PRUNE))
    hit = hits["prim"] != _abi.RT_MISS
    assert 0.3 < hit.mean() < 0.8
    tri = desc.triangles[hits["prim"][hit]]
    p = rays["o"][hit] + rays["d"][hit] * hits["t"][hit][:, None]
    plane = np.einsum("ij,ij->i", tri["n"], p - tri["a"])
    assert np.abs(plane).max() < 1e-9
    lo = np.minimum(np.minimum(tri["a"], tri["b"]), tri["c"]) - 1e-9
    hi = np.maximum(np.maximum(tri["a"], tri["b"]), tri["c"]) + 1e-9
    assert ((p >= lo) & (p <= hi)).all()
    assert (np.einsum("ij,ij->i", tri["n"], rays["d"][hit]) < 0).all()          # one-sided: only front faces
    a, sa = cam.render_image(world, spp=2, seed=8, with_stats=True)
    b = cam.render_image(world, spp=2, seed=8, max_paths_in_flight=1920 * 1080)
    assert np.array_equal(a, b)
    halves = [cam.render_image(world, spp=1, seed=8, sample_offset=r, sample_stride=2) for r in range(2)]
    np.testing.assert_allclose((halves[0] + halves[1]) / 2.0, a, rtol=1e-12, atol=1e-15)
    assert np.isfinite(a).all() and a.min() >= 0.0 and a.max() <= 1.0
    assert int(sa["samples"]) == 2 * 1920 * 1080 and int(sa["ray_segments"]) > int(sa["samples"])


def degenerate_rays(desc, seed=4):
    """Rays that stress the conservative f32 box test: exactly axis-parallel directions, origins on
    box / triangle planes, directions with denormal-small components, far and huge origins."""
    k = np.arange(4000)
    base = random_rays(len(k), seed)
    o, d = base[:, :3].copy(), base[:, 3:].copy()
    axis = np.eye(3)[k % 3] * np.where((k // 3) % 2 == 0, 1.0, -1.0)[:, None]
    sel = k % 8
    d[sel == 0] = axis[sel == 0]                                   # exactly axis-parallel (two zero components)
    d[sel == 1] = axis[sel == 1] + 1e-18 * d[sel == 1]             # tiny off-axis components
    d[sel == 2] = axis[sel == 2] + 1e-300 * d[sel == 2]            # reciprocal overflows f32 (and nearly f64)
    o[sel == 3, 2] = -0.2                                          # on the ground plane ...
    d[sel == 3, 2] = 0.0                                           # ... travelling inside it
    far = sel == 4
    nd = d[far] / np.linalg.norm(d[far], axis=1, keepdims=True)
    o[far] = o[far] * 0.3 - nd * 1e6                               # a million units away, aimed at the scene
    huge = sel == 5
    o[huge] = o[huge] * 1e9
    if len(desc.triangles):
        tri = desc.triangles[k[sel == 6] % len(desc.triangles)]
        inside = tri["a"] + 0.25 * (tri["b"] - tri["a"]) + 0.5 * (tri["c"] - tri["a"])
        o[sel == 6] = inside + np.array([0.0, 0.0, 1.0])            # straight above an interior point
        d[sel == 6] = (0.0, 0.0, -1.0)
    if len(desc.spheres):
        sp = desc.spheres[k[sel == 7] % len(desc.spheres)]
        o[sel == 7] = sp["center"] + np.array([0.0, 0.0, 1.0]) * (sp["radius"][:, None] + 2.0)
        d[sel == 7] = (0.0, 0.0, -1.0)                             # through the pole: |z'| ~ 1 (NaN-uv edge)
    return np.concatenate([o, d], -1)


def test_degenerate_rays_keep_bit_exact_hits(cover, mixed, oracle):
    for desc, world, orc in (cover, mixed):
        rays = degenerate_rays(desc)
        assert_hits_equal(world.hit(rays), orc.hit(rays, mode=oracle.BRUTE))
        assert_hits_equal(world.hit(rays, 0.0, 50.0), orc.hit(rays, 0.0, 50.0, mode=oracle.BRUTE))


def test_exact_ties_keep_t_bit_exact(cover, oracle):
    """Rays through a vertex shared by both ground triangles: the winner among equal-t primitives is
    traversal-order dependent (the contract's stated exception, src/hittable.rs:142 half-open range);
    t itself must still be bit-exact and the primitive must be one of the tied ones."""
    desc, world, orc = cover
    tri = desc.triangles
    o = np.concatenate([tri["a"], tri["c"]]) + np.array([0.0, 0.0, 1.0])
    rays = np.concatenate([o, np.tile([0.0, 0.0, -1.0], (len(o), 1))], -1)
    g, r = world.hit(rays), orc.hit(rays, mode=oracle.BRUTE)
    assert np.array_equal(g["t"], r["t"]) and (g["t"] == 1.0).all()
    assert (g["prim"] >= len(desc.spheres)).all()


def test_node_formats_agree(monkeypatch, oracle):
    """RT_HIP_NODE_BOXES selects exact f64 boxes, conservative f32 boxes, or f32 boxes quantised to a
    16-bit grid, as BVH2 (q16) or collapsed to a BVH4 (w4: default for the per-lane traversal when the
    grid is fine enough), or the BVH8 with 8-bit boxes on node-local grids and octant-ordered children (w8: its own walk
    kernel, its own record pool shared with the camera-ray kernel's BVH2).  Boxes only cull: all formats must give the
    same hits and the same image, bit for bit."""
    desc = scenes.mixed_test_scene(seed=5, n_spheres=60, quads=40)
    worlds = {}
    for fmt in ("w8", "w4", "w4f", "q16", "f32", "f64"):
        monkeypatch.setenv("RT_HIP_NODE_BOXES", fmt)
        worlds[fmt] = World.build(desc)
    monkeypatch.delenv("RT_HIP_NODE_BOXES")
    worlds["default"] = World.build(desc)
    assert [worlds[f].info()["node_bytes"] for f in ("w8", "w4", "w4f", "q16", "f32", "f64")] == [80, 64, 128, 32, 64, 128]
    assert [worlds[f].info()["bvh_width"] for f in ("default", "w8", "w4", "w4f", "q16", "f32", "f64")] == [4, 8, 4, 4, 2, 2, 2]
    rays = np.concatenate([random_rays(40_000, 17), degenerate_rays(desc, 6)])
    hits = {f: w.hit(rays) for f, w in worlds.items()}
    assert_hits_equal(hits["f32"], hits["f64"])
    assert_hits_equal(hits["q16"], hits["f64"])
    assert_hits_equal(hits["w4"], hits["f64"])
    assert_hits_equal(hits["w4f"], hits["f64"])
    assert_hits_equal(hits["w8"], hits["f64"])
    assert_hits_equal(hits["default"], hits["f64"])
    assert_hits_equal(hits["q16"], oracle.OracleScene(desc).hit(rays, mode=oracle.BVH_PRUNE))
    # on ordinary rays the wider margins / coarser boxes accept only a few extra nodes
    nv = {f: int(w.hit(rays[:40_000], with_stats=True)[1]["nodes_visited"]) for f, w in worlds.items()}
    assert nv["f64"] <= nv["f32"] <= nv["f64"] * 1.01 and nv["f32"] <= nv["q16"] <= nv["f32"] * 1.10
    assert nv["w4"] < 0.7 * nv["q16"]                      # a BVH4 visit replaces about two BVH2 visits
    assert nv["w8"] < 0.85 * nv["w4"]                      # and a BVH8 visit about 1.3 BVH4 visits (octant order, no sort)
    cam = Camera.new((9.0, -7.0, 4.0), (0.0, 0.0, 0.8), (0.0, 0.0, 1.0), 10.0, 0.3, 96, 64, 4, 12, 35.0)
    img = cam.render_image(worlds["f64"], spp=3, seed=2)
    assert np.array_equal(cam.render_image(worlds["f32"], spp=3, seed=2), img)
    assert np.array_equal(cam.render_image(worlds["q16"], spp=3, seed=2), img)
    assert np.array_equal(cam.render_image(worlds["w4"], spp=3, seed=2), img)
    assert np.array_equal(cam.render_image(worlds["w4f"], spp=3, seed=2), img)
    assert np.array_equal(cam.render_image(worlds["w8"], spp=3, seed=2), img)


def test_coarse_grid_is_rejected_by_default_and_still_exact_when_forced(monkeypatch, cover, oracle):
    """The cover scene's 10000-unit ground plane makes the 16-bit grid coarser than the spheres: the
    library keeps f32 boxes there (as a BVH4 with 128-byte nodes); forcing the grid must still be conservative
    (same hits)."""
    desc, world, orc = cover
    assert world.info()["node_bytes"] == 128 and world.info()["bvh_width"] == 4
    monkeypatch.setenv("RT_HIP_NODE_BOXES", "q16")
    forced = World.build(desc)
    monkeypatch.delenv("RT_HIP_NODE_BOXES")
    assert forced.info()["node_bytes"] == 32
    rays = np.concatenate([random_rays(30_000, 23), degenerate_rays(desc, 8)])
    ref = orc.hit(rays, mode=oracle.BRUTE)
    assert_hits_equal(forced.hit(rays), ref)
    monkeypatch.setenv("RT_HIP_NODE_BOXES", "w4")
    forced4 = World.build(desc)
    monkeypatch.delenv("RT_HIP_NODE_BOXES")
    assert forced4.info()["bvh_width"] == 4
    assert_hits_equal(forced4.hit(rays), ref)
    # the BVH8's grids are local to each node: the huge ground plane costs it nothing (spheres, a few triangles,
    # defocus camera: the whole render path on the other walk)
    monkeypatch.setenv("RT_HIP_NODE_BOXES", "w8")
    w8 = World.build(desc)
    monkeypatch.delenv("RT_HIP_NODE_BOXES")
    assert w8.info()["bvh_width"] == 8
    assert_hits_equal(w8.hit(rays), ref)
    cam = scenes.cover_camera(128, 72, spp=4, max_depth=50, defocus=True)
    assert np.array_equal(cam.render_image(w8, spp=4, seed=6), cam.render_image(world, spp=4, seed=6))


def test_converged_image_relmse_at_4096_spp(cover, oracle):
    """north_star's image criterion: relative MSE < 1e-3 at 4096 spp against the (counter-RNG)
    reference path.  With shared RNG the two renders follow the same paths, so the error is at
    rounding level; the criterion is asserted as stated and the actual value far below it."""
    desc, world, orc = cover
    cam = scenes.cover_camera(48, 27, spp=4096, max_depth=50, defocus=True)
    img = cam.render_image(world, spp=4096, seed=11)
    ref = orc.render(cam, 4096, seed=11, mode=oracle.BVH_PRUNE, nthreads=16)
    relmse = float(np.mean((img - ref) ** 2 / (ref ** 2 + 1e-4)))
    assert relmse < 1e-3 and relmse < 1e-10, relmse
    # and the estimate itself has converged: an independent seed agrees to the same criterion
    other = cam.render_image(world, spp=4096, seed=12)
    assert float(np.mean((other - ref) ** 2 / (ref ** 2 + 1e-4))) < 1e-2


def test_persistent_refill_kernel_is_bit_exact(monkeypatch, cover, mixed, oracle):
    """The persistent traversal kernel (finished lanes take new rays) against the oracle and against the
    one-chunk-per-wave kernel: a few persistent waves over many ragged chunks, then a render through it."""
    desc, world, orc = mixed
    rays = random_rays(30_011, 77)                       # 469 chunks, the last one ragged
    monkeypatch.setenv("RT_HIP_NO_REFILL", "1")
    plain = world.hit(rays)
    monkeypatch.delenv("RT_HIP_NO_REFILL")
    for slots in (1, 7, 64):
        monkeypatch.setenv("RT_HIP_REFILL_SLOTS", str(slots))
        got = world.hit(rays)
        assert np.array_equal(got["prim"], plain["prim"]) and np.array_equal(got["t"], plain["t"])
    assert_hits_equal(got, orc.hit(rays, mode=oracle.BRUTE))
    desc, world, orc = cover
    cam = scenes.cover_camera(128, 72, spp=3, max_depth=50, defocus=True)
    monkeypatch.setenv("RT_HIP_REFILL_SLOTS", "5")
    img, st = cam.render_image(world, spp=3, seed=12345, with_stats=True)
    monkeypatch.setenv("RT_HIP_NO_REFILL", "1")
    img0, st0 = cam.render_image(world, spp=3, seed=12345, with_stats=True)
    assert np.array_equal(img, img0)
    for k in ("ray_segments", "nodes_visited", "prims_tested"):
        assert int(st[k]) == int(st0[k])


def test_sample_grouping_of_camera_waves_does_not_change_the_image(monkeypatch, cover):
    """RT_HIP_GROUP: how many samples of a pixel share a wave of camera rays (path-id layout only).  RNG keys
    and the per-pixel sum order do not depend on it, so every grouping gives the same image bit for bit."""
    desc, world, _ = cover
    cam = scenes.cover_camera(96, 64, spp=64, max_depth=12, defocus=True)
    imgs = {}
    for g in (1, 4, 16, 64):
        monkeypatch.setenv("RT_HIP_GROUP", str(g))
        imgs[g], st = cam.render_image(world, spp=64, seed=9, with_stats=True)
        assert int(st["samples"]) == 96 * 64 * 64
    for g in (4, 16, 64):
        assert np.array_equal(imgs[g], imgs[1]), g
    monkeypatch.setenv("RT_HIP_GROUP", "16")                 # ragged: 50 samples -> passes fall back to smaller groups
    odd = scenes.cover_camera(70, 38, spp=50, max_depth=12, defocus=True)
    a = odd.render_image(world, spp=50, seed=2)
    monkeypatch.setenv("RT_HIP_GROUP", "1")
    assert np.array_equal(a, odd.render_image(world, spp=50, seed=2))


def test_obj_import_to_closest_hits(tmp_path, oracle):
    """f3 end to end: an OBJ file -> load_obj -> World.build -> closest hits, against the oracle on the same
    flattened scene."""
    from rt_amd import importers
    rng = np.random.default_rng(4)
    n = 24
    gx, gy = np.meshgrid(np.linspace(-4, 4, n), np.linspace(-4, 4, n), indexing="ij")
    gz = 0.6 * np.sin(gx) * np.cos(1.3 * gy) + 0.05 * rng.standard_normal(gx.shape)
    lines = ["o terrain"] + [f"v {x:.6f} {y:.6f} {z:.6f}" for x, y, z in zip(gx.ravel(), gy.ravel(), gz.ravel())]
    for i in range(n - 1):
        for j in range(n - 1):
            v = i * n + j + 1
            lines.append(f"f {v} {v + n} {v + n + 1} {v + 1}")            # quads: fan-triangulated by the loader
    path = tmp_path / "terrain.obj"
    path.write_text("\n".join(lines) + "\n")
    soups = importers.load_obj(str(path), Lambertian.new_rgb_solid(0.6, 0.5, 0.4))
    assert len(soups) == 1 and len(soups[0]) == 2 * (n - 1) * (n - 1)
    desc = SceneDesc.from_shapes(soups + [Sphere((0.0, 0.0, 2.0), 0.7, Lambertian.new_rgb_solid(0.2, 0.2, 0.8))])
    world, orc = World.build(desc), oracle.OracleScene(desc)
    rays = random_rays(20_000, 41, -4.0, 4.0)
    gpu = world.hit(rays)
    assert_hits_equal(gpu, orc.hit(rays, mode=oracle.BRUTE))
    assert (gpu["prim"] != _abi.RT_MISS).mean() > 0.2


def test_device_bvh_builder_gives_the_same_hits(monkeypatch, oracle):
    """f1: RT_HIP_BUILDER=lbvh builds the tree on the device (Morton order, Karras radix tree, bottom-up fit).
    Another tree, the same closest hits and the same image."""
    scenes_ = {"mixed": scenes.mixed_test_scene(seed=5, n_spheres=60, quads=40), "heightfield": scenes.heightfield_scene(quads=96)}
    for name, desc in scenes_.items():
        host = World.build(desc)
        monkeypatch.setenv("RT_HIP_BUILDER", "lbvh")
        dev = World.build(desc)
        monkeypatch.delenv("RT_HIP_BUILDER")
        hi, di = host.info(), dev.info()
        assert (hi["builder"], di["builder"]) == (0, 1), name
        assert di["bvh_depth"] <= 32 and di["max_leaf_size"] <= 4 and di["n_prims"] == hi["n_prims"]
        lo, hi_ = (-45.0, 45.0) if name == "heightfield" else (-8.0, 8.0)
        rays = np.concatenate([random_rays(30_000, 51, lo, hi_), degenerate_rays(desc, 3)])
        got = dev.hit(rays)
        assert_hits_equal(got, host.hit(rays))
        assert_hits_equal(got, oracle.OracleScene(desc).hit(rays, mode=oracle.BVH_PRUNE))
        cam = scenes.heightfield_camera(160, 90) if name == "heightfield" else Camera.new((9.0, -7.0, 4.0), (0.0, 0.0, 0.8), (0.0, 0.0, 1.0), 10.0, 0.3, 96, 64, 4, 12, 35.0)
        assert np.array_equal(cam.render_image(dev, spp=4, seed=2), cam.render_image(host, spp=4, seed=2)), name
    monkeypatch.setenv("RT_HIP_BUILDER", "lbvh")                      # tiny scenes are declined: host builder
    tiny = World.build(SceneDesc.from_shapes([Sphere((0.0, 0.0, 0.0), 1.0, Lambertian.new_rgb_solid(0.5, 0.5, 0.5))]))
    assert tiny.info()["builder"] == 0


def test_degenerate_scenes_and_rays_terminate_and_match(monkeypatch, oracle):
    """Coincident primitives (every split is degenerate: median fallback, deep trees), zero-area triangles and rays
    with a zero / NaN / infinite direction: no hang, no fault, and the same answers as the oracle (t only where
    coincident primitives tie)."""
    mat = Lambertian.new_rgb_solid(0.5, 0.5, 0.5)
    tri = lambda z: Triangle((-1.0, -1.0, z), (1.0, -1.0, z), (0.0, 1.0, z), mat)
    shapes = [tri(0.0) for _ in range(3000)] + [Sphere((0.0, 0.0, 2.0), 0.5, mat) for _ in range(500)]
    shapes += [Triangle((0.0, 0.0, 1.0), (0.0, 0.0, 1.0), (0.0, 0.0, 1.0), mat),          # a point
               Triangle((0.0, 0.0, 1.0), (1.0, 1.0, 1.0), (2.0, 2.0, 1.0), mat)]          # a segment
    desc = SceneDesc.from_shapes(shapes)
    orc = oracle.OracleScene(desc)
    rays = random_rays(4000, 61, -1.5, 1.5)
    rays[:, 2] = 4.0
    rays[:, 5] = -np.abs(rays[:, 5]) - 0.2
    weird = np.array([[0, 0, 4, 0, 0, 0], [0, 0, 4, np.nan, 0, -1], [0, 0, 4, 0, 0, -np.inf], [np.nan, 0, 4, 0, 0, -1],
                      [0, 0, 4, 0, 0, -1], [0.25, 0.1, 4, 0, 0, -1]], dtype=np.float64)
    rays = np.concatenate([rays, weird])
    ref = orc.hit(rays, mode=oracle.BRUTE)
    for builder in ("sah", "lbvh"):
        monkeypatch.setenv("RT_HIP_BUILDER", builder)
        world = World.build(desc)
        assert world.info()["bvh_depth"] <= 32
        got = world.hit(rays)
        assert np.array_equal(got["t"], ref["t"]), builder
        assert np.array_equal(got["prim"] == _abi.RT_MISS, ref["prim"] == _abi.RT_MISS), builder
        cam = Camera.new((0.0, -6.0, 3.0), (0.0, 0.0, 0.5), (0.0, 0.0, 1.0), 6.0, 0.0, 64, 48, 2, 8, 40.0)
        img = cam.render_image(world, spp=2, seed=1)
        assert np.isfinite(img).all()
