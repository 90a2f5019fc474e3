"""Parity exchange with the unmodified reference: the route to pin closest-hit parity on a machine with cargo.

The reference cannot be built in this repository's image (no Rust toolchain), so parity is pinned against the CPU
restatement only.  This module writes a scene + ray file that `integration/rust/dump_hits.rs` (an example program
for the reference crate, source only) reads, building the shapes through the reference's own constructors
(`Sphere::new_facing`, `Triangle::new_with_uv`: src/hittable.rs:283-297, :184-208) and dumping `World::hit`
(src/hittable.rs:137-148) as `(prim, t bits)`; `compare` runs the same rays through `rt_trace_closest` and checks
the dump record by record.

  python -m rt_amd.hits_exchange write cover 320 180 scene.rthx        # scene name from rt_amd.scenes, image size
  cargo run --release --example dump_hits -- scene.rthx hits.rthd      # in the reference checkout
  python -m rt_amd.hits_exchange compare scene.rthx hits.rthd          # needs the GPU library

With `integration/rust/patches/rt-counter-rng.patch` applied to the reference (its six thread_rng() draws become
slots of the Philox counter this library uses), `integration/rust/dump_render.rs` dumps a whole render and

  python -m rt_amd.hits_exchange compare-render scene.rthx render.rtrd  # needs the GPU library

renders the same scene, camera, seed and sky state through `rt_render` and compares the per-pixel means.

Formats are little-endian and documented in dump_hits.rs / dump_render.rs.  The rays are the un-jittered pixel-centre rays of the
scene's camera (`Camera::debug_ray`, src/camera.rs:256-260), un-normalised as the reference hands them to Ray::new.
"""
import struct
import sys

import numpy as np

from . import _abi

MAGIC_X = b"RTHX0001"
MAGIC_D = b"RTHD0001"
MAGIC_R = b"RTRD0001"
dump_dt = np.dtype([("prim", "<u4"), ("_pad", "<u4"), ("t_bits", "<u8")])


def pixel_centre_rays(cam):
    """(W*H, 6) float64: Camera::debug_ray for every pixel, row-major."""
    r = cam.rec[0]
    xs, ys = np.meshgrid(np.arange(cam.width, dtype=np.float64), np.arange(cam.height, dtype=np.float64))
    p = (np.asarray(r["pixel00"])[None, None, :] + xs[..., None] * np.asarray(r["pixel_du"])[None, None, :]
         + ys[..., None] * np.asarray(r["pixel_dv"])[None, None, :])
    c = np.asarray(r["center"], dtype=np.float64)
    out = np.empty((cam.height, cam.width, 6), dtype=np.float64)
    out[..., :3] = c
    out[..., 3:] = p - c
    return out.reshape(-1, 6)


def write_exchange(path, desc, rays, t_min=0.001, t_max=float(np.finfo(np.float64).max)):
    rays = np.ascontiguousarray(rays, dtype="<f8").reshape(-1, 6)
    sph = np.empty((len(desc.spheres), 7), dtype="<f8")
    sph[:, 0:3], sph[:, 3], sph[:, 4:7] = desc.spheres["center"], desc.spheres["radius"], desc.spheres["face"]
    tri = np.empty((len(desc.triangles), 15), dtype="<f8")
    tri[:, 0:3], tri[:, 3:6], tri[:, 6:9] = desc.triangles["a"], desc.triangles["b"], desc.triangles["c"]
    tri[:, 9:15] = desc.triangles["uv"].reshape(-1, 6)
    with open(path, "wb") as f:
        f.write(MAGIC_X)
        f.write(struct.pack("<QQQdd", len(sph), len(tri), len(rays), t_min, t_max))
        f.write(sph.tobytes())
        f.write(tri.tobytes())
        f.write(rays.tobytes())


def read_exchange(path):
    """-> dict(spheres (n,7), triangles (n,15), rays (n,6), t_min, t_max)"""
    with open(path, "rb") as f:
        buf = f.read()
    if buf[:8] != MAGIC_X:
        raise ValueError("not an RTHX0001 file")
    ns, nt, nr, t_min, t_max = struct.unpack_from("<QQQdd", buf, 8)
    off = 8 + 40
    if len(buf) != off + 8 * (7 * ns + 15 * nt + 6 * nr):
        raise ValueError("RTHX0001 file has the wrong length")
    sph = np.frombuffer(buf, "<f8", 7 * ns, off).reshape(ns, 7)
    off += 56 * ns
    tri = np.frombuffer(buf, "<f8", 15 * nt, off).reshape(nt, 15)
    off += 120 * nt
    rays = np.frombuffer(buf, "<f8", 6 * nr, off).reshape(nr, 6)
    return {"spheres": sph, "triangles": tri, "rays": rays, "t_min": t_min, "t_max": t_max}


def write_dump(path, hits):
    """hits: hit_dt array (t, prim); misses are stored as (0xFFFFFFFF, 0)."""
    rec = np.zeros(len(hits), dump_dt)
    miss = hits["prim"] == _abi.RT_MISS
    rec["prim"] = hits["prim"]
    rec["t_bits"] = np.where(miss, np.uint64(0), np.ascontiguousarray(hits["t"]).view("<u8"))
    with open(path, "wb") as f:
        f.write(MAGIC_D)
        f.write(struct.pack("<Q", len(rec)))
        f.write(rec.tobytes())


def read_dump(path):
    with open(path, "rb") as f:
        buf = f.read()
    if buf[:8] != MAGIC_D:
        raise ValueError("not an RTHD0001 file")
    (n,) = struct.unpack_from("<Q", buf, 8)
    if len(buf) != 16 + 16 * n:
        raise ValueError("RTHD0001 file has the wrong length")
    return np.frombuffer(buf, dump_dt, n, 16)


def write_render_dump(path, image, spp, max_depth, seed, camera12, sky):
    """image (H, W, 3) f64; camera12 = center, lookat, up, vfov, focus_distance, defocus_angle; sky = sky_dt record."""
    h, w, _ = image.shape
    with open(path, "wb") as f:
        f.write(MAGIC_R)
        f.write(struct.pack("<IIIIQ", w, h, spp, max_depth, seed))
        f.write(np.asarray(camera12, "<f8").tobytes())
        f.write(np.asarray(sky["params"], "<f4").reshape(27).tobytes())
        f.write(np.asarray(sky["radiances"], "<f4").reshape(3).tobytes())
        f.write(np.asarray(sky["sun"], "<f8").reshape(3).tobytes())
        f.write(np.ascontiguousarray(image, "<f8").tobytes())


def read_render_dump(path):
    """-> dict(width, height, spp, max_depth, seed, camera (12,), sky (sky_dt record), image (H, W, 3))"""
    with open(path, "rb") as f:
        buf = f.read()
    if buf[:8] != MAGIC_R:
        raise ValueError("not an RTRD0001 file")
    w, h, spp, depth, seed = struct.unpack_from("<IIIIQ", buf, 8)
    off = 8 + 24
    if len(buf) != off + 96 + 120 + 24 + 24 * w * h:
        raise ValueError("RTRD0001 file has the wrong length")
    cam = np.frombuffer(buf, "<f8", 12, off)
    sky = _abi.one(_abi.sky_dt)
    sky["params"] = np.frombuffer(buf, "<f4", 27, off + 96)
    sky["radiances"] = np.frombuffer(buf, "<f4", 3, off + 96 + 108)
    sky["sun"] = np.frombuffer(buf, "<f8", 3, off + 96 + 120)
    image = np.frombuffer(buf, "<f8", 3 * w * h, off + 96 + 120 + 24).reshape(h, w, 3)
    return {"width": w, "height": h, "spp": spp, "max_depth": depth, "seed": seed, "camera": cam, "sky": sky,
            "image": image}


def camera_from_render_dump(rd):
    from . import api
    c = rd["camera"]
    return api.Camera.new(tuple(c[0:3]), tuple(c[3:6]), tuple(c[6:9]), float(c[10]), float(c[11]), rd["width"],
                          rd["height"], rd["spp"], rd["max_depth"], float(c[9]))


def compare_renders(ours, theirs, rtol=2e-6):
    """-> (n_values, n_bit_identical, indices beyond rtol).  Not bit-exact by nature: the sky's f32 cos/exp/pow and the
    f64 acos come from different math libraries (OCML here, the system libm under Rust: 1-2 ulp of f32, ~1e-7), the
    throughput is multiplied forward here and on the way out of the recursion there, and rayon's `.sum()` has no fixed
    association order.  2e-6 is the tolerance of every render comparison in tests/ (DESIGN.md §3)."""
    a, b = np.ascontiguousarray(ours, "<f8").ravel(), np.ascontiguousarray(theirs, "<f8").ravel()
    if a.shape != b.shape:
        raise ValueError(f"image sizes differ: {a.shape} vs {b.shape}")
    same = int(np.count_nonzero(a.view("<u8") == b.view("<u8")))
    bad = np.nonzero(~(np.abs(a - b) <= rtol * np.maximum(np.abs(a), np.abs(b))))[0]
    return a.size, same, bad


def desc_from_exchange(ex, sky=None):
    """SceneDesc of the exchange file's shapes (one grey Lambertian; normals as Triangle::new computes them)."""
    from . import api
    mat = api.Lambertian.new_rgb_solid(0.5, 0.5, 0.5)
    t = ex["triangles"]
    soup = None
    shapes = [api.Sphere.new_facing(tuple(s[0:3]), float(s[3]), mat, tuple(s[4:7])) for s in ex["spheres"]]
    if len(t):
        from .importers import TriangleSoup
        soup = TriangleSoup(t[:, 0:3].copy(), t[:, 3:6].copy(), t[:, 6:9].copy(), mat, t[:, 9:15].reshape(-1, 3, 2).copy())
        shapes.append(soup)
    return api.SceneDesc.from_shapes(shapes, sky)


def compare_dumps(ours, theirs):
    """-> (n_records, mismatching indices).  A record matches when prim and the bits of t are equal; two records that
    differ in prim but have identical t bits are ties between primitives at the same distance and are reported
    separately by the caller."""
    if len(ours) != len(theirs):
        raise ValueError(f"dump lengths differ: {len(ours)} vs {len(theirs)}")
    bad = np.nonzero((ours["prim"] != theirs["prim"]) | (ours["t_bits"] != theirs["t_bits"]))[0]
    return len(ours), bad


def _main(argv):
    if len(argv) >= 6 and argv[1] == "write":
        from . import scenes
        name, w, h, out = argv[2], int(argv[3]), int(argv[4]), argv[5]
        desc = getattr(scenes, name + "_scene")()
        cam = getattr(scenes, name + "_camera")(w, h)
        write_exchange(out, desc, pixel_centre_rays(cam), 0.001, float(cam.rec["t_max"][0]))
        print(f"{out}: {len(desc.spheres)} spheres, {len(desc.triangles)} triangles, {w * h} rays")
        return 0
    if len(argv) == 4 and argv[1] == "compare":
        from . import api
        ex = read_exchange(argv[2])
        theirs = read_dump(argv[3])
        world = api.World.build(desc_from_exchange(ex))
        hits = world.hit(ex["rays"], ex["t_min"], ex["t_max"])
        ours = np.zeros(len(hits), dump_dt)
        miss = hits["prim"] == _abi.RT_MISS
        ours["prim"] = hits["prim"]
        ours["t_bits"] = np.where(miss, np.uint64(0), np.ascontiguousarray(hits["t"]).view("<u8"))
        n, bad = compare_dumps(ours, theirs)
        ties = [i for i in bad if ours["t_bits"][i] == theirs["t_bits"][i]]
        print(f"{n} rays: {n - len(bad)} identical, {len(ties)} ties (same t bits, other primitive), "
              f"{len(bad) - len(ties)} different")
        for i in bad[:10]:
            print(f"  ray {i}: library (prim {ours['prim'][i]}, t bits {ours['t_bits'][i]:#018x})  "
                  f"reference (prim {theirs['prim'][i]}, t bits {theirs['t_bits'][i]:#018x})")
        return 0 if len(bad) == len(ties) else 1
    if len(argv) == 4 and argv[1] == "compare-render":
        from . import api
        ex = read_exchange(argv[2])
        rd = read_render_dump(argv[3])
        world = api.World.build(desc_from_exchange(ex, rd["sky"]))
        ours = camera_from_render_dump(rd).render_image(world, spp=rd["spp"], seed=rd["seed"])
        n, same, bad = compare_renders(ours, rd["image"])
        print(f"{n} values at {rd['spp']} spp: {same} bit-identical, {len(bad)} beyond rtol 2e-6")
        for i in bad[:10]:
            print(f"  value {i} (pixel {i // 3}, channel {i % 3}): library {ours.ravel()[i]!r}  reference "
                  f"{rd['image'].ravel()[i]!r}")
        return 0 if len(bad) == 0 else 1
    print(__doc__)
    return 2


if __name__ == "__main__":
    sys.exit(_main(sys.argv))
