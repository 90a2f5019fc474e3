"""GPU parity for BASELINE.json configs 4 and 5 and the code paths only they reach: image-textured TRIANGLES
(the bbox-uv tail of Triangle::hit, src/hittable.rs:469-481, handed from the traversal kernels to the shading code),
glTF -> Metal(image, Some(roughness)) (src/material.rs:20-33, src/hittable.rs:556-633), frosted dielectrics, and
the kernel variants a render goes through (coherent-wave camera kernel, fused depth-0 shading, passes in flight).
Same bar as tests/test_gpu_parity.py: (prim, t) and segment counts bit-exact, radiance within RTOL."""
import io
import json
import struct

import numpy as np
import pytest

from rt_amd import World, _abi, importers, scenes
from rt_amd.api import (Camera, CheckerTexture, Dielectric, Lambertian, SceneDesc, SolidColor, Sphere, encode_gamma8,
                        generate_ground_plane, write_ppm)

pytestmark = pytest.mark.gpu

RTOL, ATOL = 2e-6, 1e-12


def small_textured_scene():
    return scenes.textured_scene(terrain_quads=48, instances=150, subdiv=2, tex_size=64)


def textured_camera(width=160, height=90, spp=4, max_depth=50):
    return scenes.heightfield_camera(width, height, spp, max_depth)


def random_rays(n, seed, lo=-45.0, hi=45.0, z0=0.5, z1=9.0):
    k = np.arange(n)
    o = np.stack([lo + (hi - lo) * scenes.hash_u01(seed, k, c) for c in range(3)], -1)
    o[:, 2] = z0 + (z1 - z0) * scenes.hash_u01(seed, k, 7)
    d = np.stack([2.0 * scenes.hash_u01(seed + 1, k, c) - 1.0 for c in range(3)], -1)
    return np.concatenate([o, d], -1)


def assert_hits_equal(gpu, ref):
    bad = np.flatnonzero((gpu["prim"] != ref["prim"]) | (gpu["t"] != ref["t"]))
    assert bad.size == 0, f"{bad.size} of {len(ref)} hits differ, first: gpu={gpu[bad[:3]]} ref={ref[bad[:3]]}"


@pytest.fixture(scope="module")
def textured(oracle):
    desc = small_textured_scene()
    return desc, World.build(desc), oracle.OracleScene(desc)


def test_textured_triangles_closest_hits(textured, oracle):
    desc, world, orc = textured
    rays = random_rays(40_000, 3)
    gpu = world.hit(rays)
    assert_hits_equal(gpu, orc.hit(rays, mode=oracle.BRUTE))
    cam = textured_camera()
    for s in (0, 7):
        assert_hits_equal(cam.trace_primary(world, s, jitter=True), orc.trace_primary(cam, s, jitter=True, mode=oracle.BRUTE))
    nt = 2 * 48 * 48
    on_instances = (gpu["prim"] != _abi.RT_MISS) & (gpu["prim"] >= nt)
    assert on_instances.mean() > 0.03                     # the image-textured / frosted instances are really hit


@pytest.mark.parametrize("spp,seed,max_depth", [(1, 0, 50), (4, 3, 50), (3, 11, 2)])
def test_textured_triangles_render_matches_oracle(textured, oracle, spp, seed, max_depth):
    """Image texture on triangles through every hand-off of the barycentrics: the fused camera-ray kernel (depth 0),
    k_walk -> k_shade (bounces), and a low max_depth so that the depth limit is reached on textured hits."""
    desc, world, orc = textured
    cam = textured_camera(160, 90, spp, max_depth)
    img, st = cam.render_image(world, spp=spp, seed=seed, with_stats=True)
    ref, rst = orc.render(cam, spp, seed=seed, mode=oracle.BVH_PRUNE, with_stats=True)
    assert int(st["ray_segments"]) == int(rst["ray_segments"])
    np.testing.assert_allclose(img, ref, rtol=RTOL, atol=ATOL)
    assert img.std() > 0.05


def test_textured_triangle_uv_reaches_the_texel(oracle):
    """One big textured triangle pair seen head-on, Lambertian so the first hit's texel IS the pixel's attenuation:
    a wrong bu/bv hand-off (or a swapped u/v) shows up as a different image, not as noise."""
    img = scenes.procedural_image(64, 32)
    from rt_amd.api import ImageTexture, Triangle
    mat = Lambertian(ImageTexture(img))
    a, b, c, d = (-2.0, 0.0, -1.0), (2.0, 0.0, -1.0), (2.0, 0.0, 1.0), (-2.0, 0.0, 1.0)
    # wound so the stored normal faces -y (towards the camera); uvs span sub-rectangles of the image
    shapes = [Triangle.new_with_uv(a, b, c, (0.1, 0.9), (0.8, 0.9), (0.8, 0.2), mat),
              Triangle.new_with_uv(a, c, d, (0.0, 0.3), (0.9, 0.1), (0.2, 0.0), mat)]
    desc = SceneDesc.from_shapes(shapes)
    world, orc = World.build(desc), oracle.OracleScene(desc)
    cam = Camera.new((0.0, -6.0, 0.0), (0.0, 0.0, 0.0), (0.0, 0.0, 1.0), 6.0, 0.0, 96, 48, 2, 1, 30.0)
    hits = cam.trace_primary(world, 0, jitter=True)
    assert (hits["prim"] != _abi.RT_MISS).mean() > 0.3
    gpu, st = cam.render_image(world, spp=2, seed=5, with_stats=True)
    ref, rst = orc.render(cam, 2, seed=5, mode=oracle.BRUTE, with_stats=True)
    assert int(st["ray_segments"]) == int(rst["ray_segments"])
    np.testing.assert_allclose(gpu, ref, rtol=RTOL, atol=ATOL)


def make_glb(path):
    """A .glb with one mesh (bumpy 14x14-quad sheet, TEXCOORD_0) and an embedded PNG base-colour texture."""
    from PIL import Image as PILImage
    n = 15
    gx, gy = np.meshgrid(np.linspace(-3, 3, n), np.linspace(-3, 3, n), indexing="ij")
    gz = 0.8 + 0.5 * np.sin(1.1 * gx) * np.cos(0.9 * gy)
    pos = np.stack([gx, gy, gz], -1).reshape(-1, 3).astype(np.float32)
    uv = np.stack([(gx + 3) / 6, (gy + 3) / 6], -1).reshape(-1, 2).astype(np.float32)
    idx = []
    for i in range(n - 1):
        for j in range(n - 1):
            v = i * n + j
            idx += [v, v + n, v + n + 1, v, v + n + 1, v + 1]       # counter-clockwise from +z: normals face up
    idx = np.array(idx, dtype=np.uint16)
    png = io.BytesIO()
    PILImage.fromarray(scenes.procedural_image(48, 40), "RGB").save(png, "PNG")
    parts = [pos.tobytes(), idx.tobytes(), uv.tobytes(), png.getvalue()]
    parts = [p + b"\0" * (-len(p) % 4) for p in parts]
    offs = np.cumsum([0] + [len(p) for p in parts])
    blob = b"".join(parts)
    doc = {
        "asset": {"version": "2.0"},
        "buffers": [{"byteLength": len(blob)}],
        "bufferViews": [{"buffer": 0, "byteOffset": int(offs[k]), "byteLength": len(parts[k])} for k in range(4)],
        "accessors": [{"bufferView": 0, "componentType": 5126, "count": len(pos), "type": "VEC3"},
                      {"bufferView": 1, "componentType": 5123, "count": len(idx), "type": "SCALAR"},
                      {"bufferView": 2, "componentType": 5126, "count": len(uv), "type": "VEC2"}],
        "images": [{"bufferView": 3, "mimeType": "image/png"}],
        "textures": [{"source": 0}],
        "materials": [{"pbrMetallicRoughness": {"roughnessFactor": 0.35, "baseColorTexture": {"index": 0}}}],
        "meshes": [{"primitives": [{"attributes": {"POSITION": 0, "TEXCOORD_0": 2}, "indices": 1, "material": 0}]}],
        "nodes": [{"mesh": 0}],
    }
    js = json.dumps(doc).encode()
    js += b" " * (-len(js) % 4)
    body = struct.pack("<II", len(js), 0x4E4F534A) + js + struct.pack("<II", len(blob), 0x004E4942) + blob
    with open(path, "wb") as f:
        f.write(struct.pack("<4sII", b"glTF", 2, 12 + len(body)) + body)


def test_gltf_textured_mesh_to_render_vs_oracle(tmp_path, oracle):
    """f3 end to end on the GPU: .glb with an embedded PNG -> load_gltf -> Metal(image, Some(roughness)) on triangles
    -> World.build -> closest hits and render against the oracle on the same flattened scene."""
    path = str(tmp_path / "sheet.glb")
    make_glb(path)
    soups = importers.load_gltf(path)
    assert len(soups) == 1 and len(soups[0]) == 2 * 14 * 14
    m = soups[0].material
    assert m.fuzz == float(np.float32(0.35)) and m.texture.image.shape == (40, 48, 3)
    ground = Lambertian(CheckerTexture(1.5, SolidColor((0.1, 0.1, 0.1)), SolidColor((0.9, 0.9, 0.9))))
    shapes = generate_ground_plane(40.0, 40.0, -0.2, ground, True) + soups + \
        [Sphere((0.0, 0.0, 2.6), 0.6, Dielectric.new_frosted(1.5, 0.05)), Sphere((2.0, -2.0, 2.2), 0.5, Dielectric(1.5))]
    desc = SceneDesc.from_shapes(shapes)
    which, sub = desc.shape_of([0, 1, 2, 3, 4, 4 + 391])
    assert which.tolist() == [3, 4, 0, 1, 2, 2] and sub.tolist() == [0, 0, 0, 0, 0, 391]
    world, orc = World.build(desc), oracle.OracleScene(desc)
    rays = random_rays(30_000, 13, -4.0, 4.0, 0.0, 5.0)
    gpu = world.hit(rays)
    assert_hits_equal(gpu, orc.hit(rays, mode=oracle.BRUTE))
    assert ((gpu["prim"] >= 4) & (gpu["prim"] != _abi.RT_MISS)).mean() > 0.1          # the sheet is hit
    cam = Camera.new((7.0, -6.0, 5.0), (0.0, 0.0, 0.8), (0.0, 0.0, 1.0), 10.0, 0.2, 128, 80, 4, 20, 35.0)
    img, st = cam.render_image(world, spp=4, seed=21, with_stats=True)
    ref, rst = orc.render(cam, 4, seed=21, mode=oracle.BRUTE, with_stats=True)
    assert int(st["ray_segments"]) == int(rst["ray_segments"])
    np.testing.assert_allclose(img, ref, rtol=RTOL, atol=ATOL)


def test_kernel_variants_render_the_same_image(monkeypatch, textured):
    """The render path's switches change WHICH kernels run, never the result: per-lane instead of coherent-wave
    camera rays (RT_HIP_NO_PACKET), k_shade instead of fused depth-0 shading (RT_HIP_NO_FUSE), 1 / 4 / 8 passes in
    flight (RT_HIP_PIPES), the depth from which the persistent trace-and-shade kernel takes over (RT_HIP_TAIL_DEPTH:
    1 = straight after the camera rays, 1000 = never), pass staggering, one-chunk-per-wave launches.  Images and
    segment counts must be identical bit for bit."""
    desc, world, _ = textured
    cam = textured_camera(192, 108, 12, 50)
    kw = dict(spp=12, seed=17, max_paths_in_flight=192 * 108 * 2, with_stats=True)      # six passes
    base, st0 = cam.render_image(world, **kw)
    assert int(st0["passes"]) == 6
    for var, val in (("RT_HIP_NO_PACKET", "1"), ("RT_HIP_NO_FUSE", "1"), ("RT_HIP_PIPES", "1"), ("RT_HIP_PIPES", "4"),
                     ("RT_HIP_PIPES", "8"), ("RT_HIP_TAIL_DEPTH", "1"), ("RT_HIP_TAIL_DEPTH", "3"), ("RT_HIP_TAIL_DEPTH", "6"),
                     ("RT_HIP_TAIL_DEPTH", "1000"), ("RT_HIP_STAGGER", "0"), ("RT_HIP_NO_REFILL", "1")):
        monkeypatch.setenv(var, val)
        img, st = cam.render_image(world, **kw)
        monkeypatch.delenv(var)
        assert np.array_equal(img, base), (var, val)
        assert int(st["ray_segments"]) == int(st0["ray_segments"]), (var, val)
    cover = World.build(scenes.cover_scene())
    ccam = scenes.cover_camera(160, 90, spp=6, max_depth=50, defocus=True)
    ckw = dict(spp=6, seed=4, max_paths_in_flight=160 * 90 * 2)
    cbase = ccam.render_image(cover, **ckw)
    for var, val in (("RT_HIP_NO_PACKET", "1"), ("RT_HIP_NO_FUSE", "1"), ("RT_HIP_PIPES", "1"), ("RT_HIP_TAIL_DEPTH", "1"),
                     ("RT_HIP_TAIL_DEPTH", "4")):
        monkeypatch.setenv(var, val)
        assert np.array_equal(ccam.render_image(cover, **ckw), cbase), (var, val)
        monkeypatch.delenv(var)


def test_walk_kernel_modes_give_the_same_hits_and_image(monkeypatch, textured, oracle):
    """The three instances of the BVH4 walk — plain, every-lane-fetches (scenes that live in the caches:
    RT_HIP_ALL_LANES_LOAD, read when the scene is created) and overflow stack (deep trees; RT_HIP_OVF_LDS_ROWS=4 keeps
    only four stack rows in LDS, so nearly every push and pop of this small scene goes through the device-memory rows)
    — find the same (prim, t) as the oracle and render the same image bit for bit."""
    desc, world, orc = textured
    rays = random_rays(40_000, 29)
    want = orc.hit(rays, mode=oracle.BRUTE)
    cam = textured_camera(160, 90, 6, 50)
    kw = dict(spp=6, seed=3, max_paths_in_flight=160 * 90 * 2, with_stats=True)
    base, st0 = cam.render_image(world, **kw)
    assert_hits_equal(world.hit(rays), want)
    # ... and two other trees over the same primitives: the two-level build (a tree per separate mesh — the scene's
    # instances — under a top tree over their boxes: RT_HIP_GROUPS) and the BVH8 with its own walk kernel
    for var, val in (("RT_HIP_ALL_LANES_LOAD", "0"), ("RT_HIP_ALL_LANES_LOAD", "1"), ("RT_HIP_OVF_LDS_ROWS", "4"),
                     ("RT_HIP_OVF_LDS_ROWS", "12"), ("RT_HIP_GROUPS", "1"), ("RT_HIP_NODE_BOXES", "w8")):
        monkeypatch.setenv(var, val)
        w2 = World.build(desc)
        assert_hits_equal(w2.hit(rays), want)
        img, st = cam.render_image(w2, **kw)
        monkeypatch.delenv(var)
        assert np.array_equal(img, base), (var, val)
        assert int(st["ray_segments"]) == int(st0["ray_segments"]), (var, val)


def test_coherent_wave_kernel_hits_equal_per_lane_kernel(monkeypatch, textured, oracle):
    """k_trace_packet (what rt_render uses for camera rays) against k_walk (what rt_trace_primary uses) and the
    oracle, on the same camera rays: (prim, t) identical.  RT_HIP_PRIMARY_PACKET routes rt_trace_primary through it."""
    desc, world, orc = textured
    cover = World.build(scenes.cover_scene())
    for w, cam in ((world, textured_camera(256, 144)), (cover, scenes.cover_camera(256, 144, defocus=True))):
        for s in (0, 5):
            lane = cam.trace_primary(w, s, jitter=True, seed=3)
            monkeypatch.setenv("RT_HIP_PRIMARY_PACKET", "1")
            packet = cam.trace_primary(w, s, jitter=True, seed=3)
            monkeypatch.delenv("RT_HIP_PRIMARY_PACKET")
            assert_hits_equal(packet, lane)
    cam = textured_camera(256, 144)
    monkeypatch.setenv("RT_HIP_PRIMARY_PACKET", "1")
    assert_hits_equal(cam.trace_primary(world, 9, jitter=True), orc.trace_primary(cam, 9, jitter=True, mode=oracle.BVH_PRUNE))


def test_gpu_frame_through_the_output_stage(tmp_path, textured, oracle):
    """f4: GPU frame -> rt_encode_gamma8 -> write_ppm, against oracle frame -> orc_gamma8: bytes equal except where
    the two frames (equal to RTOL) straddle a rounding boundary of round(c^(1/2.2) * 255); header text per
    src/camera.rs:347-350."""
    import ctypes as C
    desc, world, orc = textured
    cam = textured_camera(96, 54, 8, 50)
    img = cam.render_image(world, spp=8, seed=2)
    ref = orc.render(cam, 8, seed=2, mode=oracle.BVH_PRUNE)
    g8 = encode_gamma8(img)
    r8 = np.zeros(ref.shape, dtype=np.uint8)
    oracle.lib().orc_gamma8(_abi.ptr(np.ascontiguousarray(ref)), ref.size, _abi.ptr(r8))
    scaled = np.clip(ref, 0.0, 1.0) ** (1.0 / 2.2) * 255.0
    near_boundary = np.abs(scaled - np.floor(scaled) - 0.5) < 1e-3
    assert near_boundary.mean() < 0.01
    assert np.array_equal(g8[~near_boundary], r8[~near_boundary])
    assert (np.abs(g8.astype(int) - r8.astype(int)) <= 1).all()
    path = tmp_path / "frame.ppm"
    write_ppm(str(path), g8)
    lines = path.read_text().split("\n")
    assert lines[0] == "P3" and lines[1] == "96 54 # width, height" and lines[2] == "255 # max color value"
    body = np.array(" ".join(lines[3:]).split(), dtype=np.int64).reshape(54, 96, 3)
    assert np.array_equal(body, g8)


def _frame_properties(desc, world, cam, hits, rays):
    hit = hits["prim"] != _abi.RT_MISS
    tri = desc.triangles[hits["prim"][hit]]
    p = rays["o"][hit] + rays["d"][hit] * hits["t"][hit][:, None]
    plane = np.einsum("ij,ij->i", tri["n"], p - tri["a"])
    assert np.abs(plane).max() < 1e-8
    lo = np.minimum(np.minimum(tri["a"], tri["b"]), tri["c"]) - 1e-8
    hi = np.maximum(np.maximum(tri["a"], tri["b"]), tri["c"]) + 1e-8
    assert ((p >= lo) & (p <= hi)).all()
    assert (np.einsum("ij,ij->i", tri["n"], rays["d"][hit]) < 0).all()          # one-sided: only front faces
    return hit


def _beat(capsys, msg):
    """A progress line past pytest's capture: the big tests are silent for a minute or more otherwise."""
    with capsys.disabled():
        print(f"\n[{msg}]", end="", flush=True)


def test_full_size_config4_textured_5m_triangles(oracle, capsys):
    """BASELINE.json configs[3] at full size: 5 001 288 triangles, eight 2048x2048 image textures on Metal with fuzz, 10 %
    frosted dielectric instances, 1920x1080.  The oracle (its own BVH, distance pruning) checks a strided subset of
    the camera rays and a set of incoherent rays bit-exactly and renders a band of rows; the whole frame is checked
    through size-independent properties."""
    _beat(capsys, "C4: generating 5.0 M triangles")
    desc = scenes.textured_scene()
    assert len(desc.triangles) == 5_001_288
    _beat(capsys, "C4: World.build")
    world = World.build(desc)
    info = world.info()
    assert info["n_prims"] == 5_001_288 and info["bvh_width"] == 4 and info["max_leaf_size"] <= 4
    cam = scenes.heightfield_camera(1920, 1080, 2, 50)
    hits, rays = cam.trace_primary(world, 11, jitter=True, with_rays=True)
    _beat(capsys, "C4: oracle BVH")
    orc = oracle.OracleScene(desc)
    _beat(capsys, "C4: comparing")
    sub = np.arange(0, len(rays), 41)[:50_000]
    assert_hits_equal(world.hit(rays[sub]), orc.hit(rays[sub], mode=oracle.BVH_PRUNE))
    inc = random_rays(30_000, 5)
    g = world.hit(inc)
    assert_hits_equal(g, orc.hit(inc, mode=oracle.BVH_PRUNE))
    nt = 2 * 762 * 762
    assert ((g["prim"] != _abi.RT_MISS) & (g["prim"] >= nt)).mean() > 0.02
    hit = _frame_properties(desc, world, cam, hits, rays)
    assert 0.3 < hit.mean() < 0.9
    a, sa = cam.render_image(world, spp=2, seed=8, with_stats=True)
    b = cam.render_image(world, spp=2, seed=8, max_paths_in_flight=1920 * 1080)
    assert np.array_equal(a, b)
    halves = [cam.render_image(world, spp=1, seed=8, sample_offset=r, sample_stride=2) for r in range(2)]
    np.testing.assert_allclose((halves[0] + halves[1]) / 2.0, a, rtol=1e-12, atol=1e-15)
    assert np.isfinite(a).all() and a.min() >= 0.0 and a.max() <= 1.0
    assert int(sa["samples"]) == 2 * 1920 * 1080 and int(sa["ray_segments"]) > int(sa["samples"])
    # a band of rows of the same frame against the oracle's render (same RNG keys: same paths)
    rows = (520, 528)
    ref, rst = orc.render(cam, 2, seed=8, mode=oracle.BVH_PRUNE, nthreads=16, rows=rows, with_stats=True)
    np.testing.assert_allclose(a[rows[0]:rows[1]], ref[rows[0]:rows[1]], rtol=RTOL, atol=ATOL)


def test_config5_sized_20m_triangles_4k(oracle, capsys):
    """BASELINE.json configs[4] sized (its 8-GPU run is the driver's; 'many emissives' has no reference semantics,
    SURVEY §0.1: the scene is sky-lit): 20.0 M triangles, 3840x2160, max depth 100, one sample partition of two.  Strided
    camera rays and incoherent rays bit-exact against the oracle, frame properties, partition invariance."""
    _beat(capsys, "C5: generating 20.0 M triangles")
    desc = scenes.textured_scene(terrain_quads=1414, instances=12500, tex_size=1024)
    assert len(desc.triangles) == 2 * 1414 * 1414 + 12500 * 1280
    _beat(capsys, "C5: World.build")
    world = World.build(desc)
    assert world.info()["bvh_depth"] <= 32
    cam = scenes.heightfield_camera(3840, 2160, 2, 100)
    hits, rays = cam.trace_primary(world, 3, jitter=True, with_rays=True)
    _beat(capsys, "C5: oracle BVH (about a minute)")
    orc = oracle.OracleScene(desc)
    _beat(capsys, "C5: comparing")
    sub = np.arange(0, len(rays), 331)[:25_000]
    assert_hits_equal(world.hit(rays[sub]), orc.hit(rays[sub], mode=oracle.BVH_PRUNE))
    inc = random_rays(20_000, 9)
    assert_hits_equal(world.hit(inc), orc.hit(inc, mode=oracle.BVH_PRUNE))
    hit = _frame_properties(desc, world, cam, hits, rays)
    assert 0.3 < hit.mean() < 0.9
    a, sa = cam.render_image(world, spp=2, seed=1, with_stats=True)
    halves = [cam.render_image(world, spp=1, seed=1, sample_offset=r, sample_stride=2) for r in range(2)]
    np.testing.assert_allclose((halves[0] + halves[1]) / 2.0, a, rtol=1e-12, atol=1e-15)
    assert np.isfinite(a).all() and a.min() >= 0.0 and a.max() <= 1.0
    assert int(sa["samples"]) == 2 * 3840 * 2160 and int(sa["ray_segments"]) > int(sa["samples"])
    rows = (1080, 1084)
    ref = orc.render(cam, 2, seed=1, mode=oracle.BVH_PRUNE, nthreads=16, rows=rows)
    np.testing.assert_allclose(a[rows[0]:rows[1]], ref[rows[0]:rows[1]], rtol=RTOL, atol=ATOL)
