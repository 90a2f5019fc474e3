"""Seeded synthetic scenes and the reference's cameras.

The reference's scene builders (src/scenes.rs) use an unseeded thread_rng and author-local asset
paths, so they are not reproducible; these generators mirror their *distribution* with a seeded
integer-hash RNG and emit the flattened arrays directly (SceneDesc), the way a Rust host would after
`World::build`'s flattening.  Cameras restate src/scenes.rs:17-125 (WIDTH/HEIGHT are parameters).
"""
import math

import numpy as np

from . import _abi
from .api import (Camera, CheckerTexture, Dielectric, ImageTexture, Lambertian, Metal, SceneDesc,
                  SolidColor, Sphere, generate_ground_plane, triangle_normals)

MAX_DEPTH = 100     # src/scenes.rs:15
U64 = np.uint64


# ---------------------------------------------------------------------------------------------
# splitmix64-based counter RNG for scene generation (host only; unrelated to the render RNG)
# ---------------------------------------------------------------------------------------------
def _mix(x):
    x = np.asarray(x, dtype=U64)
    with np.errstate(over="ignore"):
        x = (x + U64(0x9E3779B97F4A7C15))
        x = (x ^ (x >> U64(30))) * U64(0xBF58476D1CE4E5B9)
        x = (x ^ (x >> U64(27))) * U64(0x94D049BB133111EB)
        x = x ^ (x >> U64(31))
    return x


def hash_u01(seed, *coords):
    """Uniform [0,1) doubles from a 64-bit hash of (seed, coords...); coords broadcast."""
    h = _mix(np.asarray(seed, dtype=U64))
    for c in coords:
        with np.errstate(over="ignore"):
            h = _mix(h ^ (np.asarray(c).astype(np.int64).astype(U64) * U64(0xD6E8FEB86659FD93)))
    return (h >> U64(11)).astype(np.float64) * 2.0 ** -53


class _Stream:
    """Sequential draws (for cover_scene's loop)."""

    def __init__(self, seed):
        self.seed, self.n = seed, 0

    def u(self):
        self.n += 1
        return float(hash_u01(self.seed, self.n))

    def range(self, lo, hi):
        return lo + (hi - lo) * self.u()


# ---------------------------------------------------------------------------------------------
# cameras (src/scenes.rs:17-125)
# ---------------------------------------------------------------------------------------------
def _dist(a, b):
    return math.sqrt(sum((x - y) ** 2 for x, y in zip(a, b)))


def cam1(width=800, height=600, spp=32, max_depth=MAX_DEPTH):
    c, l = (3.0, -5.0, 0.6), (0.0, 0.0, 0.0)
    return Camera.new(c, l, (0.0, 0.0, 1.0), _dist(c, l), 0.0, width, height, spp, max_depth, 20.0)


def cam2(width=800, height=600, spp=32, max_depth=MAX_DEPTH, defocus_angle=0.7):
    return Camera.new((14.0, 3.0, 10.0), (0.0, 0.0, 0.0), (0.0, 0.0, 1.0), 16.0, defocus_angle, width, height,
                      spp, max_depth, 20.0)


def widecam(width=800, height=600, spp=32, max_depth=MAX_DEPTH):
    c, l = (-14.0, -10.0, 7.0), (0.0, 0.0, 5.0)
    return Camera.new(c, l, (0.0, 0.0, 1.0), _dist(c, l), 0.0, width, height, spp, max_depth, 40.0)


def topdown_cam(width=800, height=600, spp=32, max_depth=MAX_DEPTH):
    c, l = (0.1, 0.1, 20.0), (0.0, 0.0, 0.0)
    return Camera.new(c, l, (0.0, 0.0, 1.0), _dist(c, l), 0.7, width, height, spp, max_depth, 20.0)


# ---------------------------------------------------------------------------------------------
# S1 "cover": cover_scene(11, 11, cam, z) + checker ground (src/scenes.rs:146-238, src/main.rs:31-47)
# ---------------------------------------------------------------------------------------------
def procedural_image(w=642, h=336):
    """Seed-free stand-in for the embedded saul.webp: RGB gradient with an 8x8 checker overlay."""
    y, x = np.mgrid[0:h, 0:w]
    chk = (((x // 8) + (y // 8)) & 1).astype(np.float64)
    r = 255.0 * x / (w - 1)
    g = 255.0 * y / (h - 1)
    b = 64.0 + 128.0 * chk
    img = np.stack([r, g, b], axis=-1)
    return np.clip(np.rint(img), 0, 255).astype(np.uint8)


def cover_scene(grid_i=11, grid_j=11, camera_center=(14.0, 3.0, 10.0), z=-0.2, seed=0x52544F57):
    """S1 of SURVEY.md §8d: ground plane + 3 big spheres + <= (2*grid_i)*(2*grid_j) small spheres."""
    rng = _Stream(seed)
    shapes = []
    checker = CheckerTexture(3.0, SolidColor((0.1, 0.1, 0.1)), SolidColor((0.95, 0.95, 0.95)))
    checker_mat = Lambertian(checker)
    shapes += generate_ground_plane(10000.0, 10000.0, z, checker_mat, True)        # src/main.rs:40-49

    saul_mat = Lambertian(ImageTexture(procedural_image()))
    glass = Dielectric(1.5)
    metal = Metal.new_solid((0.7, 0.6, 0.5), None)
    R = 0.7
    saul_loc, p1, p2 = (-1.0, 1.732, R + z), (-1.0, -1.732, R + z), (2.0, 0.0, R + z)
    p3, p4, p5 = (-2.0, 0.0, R + z), (1.0, 1.732, R + z), (1.0, -1.732, R + z)
    shapes.append(Sphere(p1, R, glass))                                             # :188
    shapes.append(Sphere(p3, R, metal))                                             # :190
    shapes.append(Sphere.new_facing(saul_loc, R, saul_mat, camera_center))          # :194-196
    for i in range(-grid_i, grid_i):
        for j in range(-grid_j, grid_j):
            radius = 0.2
            albedo = (rng.u(), rng.u(), rng.u())
            off = (rng.range(0.0, 0.9), rng.range(0.0, 0.9), z)
            center = (i * 1.0 + off[0], j * 1.0 + off[1], radius + off[2])
            collide = radius + R + 1.0
            if any(_dist(center, p) < collide for p in (p1, p2, p3, p4, saul_loc, p5)):
                continue
            choose = rng.u()
            if choose > 0.95:
                mat = Dielectric(1.5)
            elif choose > 0.8:
                mat = Metal.new_solid(albedo, rng.range(0.0, 0.5))
            else:
                mat = Lambertian(SolidColor(albedo))
            shapes.append(Sphere(center, radius, mat))
    return SceneDesc.from_shapes(shapes)


def cover_camera(width, height, spp=64, max_depth=50, defocus=True):
    """cam2 geometry (src/scenes.rs:44-68); defocus=False gives the RNG-free primary-ray parity camera."""
    return cam2(width, height, spp, max_depth, 0.7 if defocus else 0.0)


# ---------------------------------------------------------------------------------------------
# S3: procedural height-field mesh (~1M triangles at the default size)
# ---------------------------------------------------------------------------------------------
def _value_noise(seed, x, y):
    """Bilinear value noise on the integer lattice from the 64-bit hash (smoothstep fade)."""
    x0, y0 = np.floor(x), np.floor(y)
    fx, fy = x - x0, y - y0
    sx, sy = fx * fx * (3.0 - 2.0 * fx), fy * fy * (3.0 - 2.0 * fy)
    ix, iy = x0.astype(np.int64), y0.astype(np.int64)
    v00 = hash_u01(seed, ix, iy)
    v10 = hash_u01(seed, ix + 1, iy)
    v01 = hash_u01(seed, ix, iy + 1)
    v11 = hash_u01(seed, ix + 1, iy + 1)
    return (v00 * (1 - sx) + v10 * sx) * (1 - sy) + (v01 * (1 - sx) + v11 * sx) * sy


def heightfield_vertices(quads, extent=50.0, amplitude=5.0, seed=3, octaves=4, base_cells=8.0):
    g = np.linspace(-extent, extent, quads + 1)
    X, Y = np.meshgrid(g, g, indexing="xy")
    u, v = (X + extent) / (2 * extent) * base_cells, (Y + extent) / (2 * extent) * base_cells
    hgt, amp, tot = np.zeros_like(X), 1.0, 0.0
    for o in range(octaves):
        hgt += amp * _value_noise(seed + 1000 * o, u * 2 ** o, v * 2 ** o)
        tot += amp
        amp *= 0.5
    return X, Y, amplitude * hgt / tot


def _tri_records(a, b, c, material_ids):
    T = np.zeros(len(a), _abi.triangle_dt)
    T["a"], T["b"], T["c"] = a, b, c
    T["uv"] = np.array([[0.0, 0.0], [1.0, 0.0], [0.5, 1.0]])       # Triangle::new defaults (src/hittable.rs:175-177)
    T["n"] = triangle_normals(a, b, c)
    T["material"] = material_ids
    return T


def heightfield_scene(quads=708, extent=50.0, amplitude=5.0, seed=3, patch=16):
    """S3 of SURVEY.md §8d: quads x quads height field = 2*quads^2 triangles wound so the stored
    normal faces +Z (triangles are one-sided, src/hittable.rs:428); one material per patch x patch
    quads: Lambertian(solid U[0.2,0.9]^3) or Metal(solid U[0.5,0.95]^3, Some(U[0,0.3)))."""
    X, Y, Z = heightfield_vertices(quads, extent, amplitude, seed)
    P = np.stack([X, Y, Z], axis=-1)                               # (q+1, q+1, 3), [row=y][col=x]
    p00, p10 = P[:-1, :-1].reshape(-1, 3), P[:-1, 1:].reshape(-1, 3)
    p01, p11 = P[1:, :-1].reshape(-1, 3), P[1:, 1:].reshape(-1, 3)
    # counter-clockwise seen from +Z  =>  normal +Z
    a = np.concatenate([p00, p00])
    b = np.concatenate([p10, p11])
    c = np.concatenate([p11, p01])
    qy, qx = np.divmod(np.arange(quads * quads), quads)
    npatch = (quads + patch - 1) // patch
    pid = (qy // patch) * npatch + (qx // patch)
    mat_ids = np.concatenate([pid, pid]).astype(np.uint32)
    nm = npatch * npatch
    k = np.arange(nm)
    is_lambert = (hash_u01(seed + 77, k, 0) < 0.5)
    col = np.stack([hash_u01(seed + 78, k, c_) for c_ in range(3)], axis=-1)
    fuzz = hash_u01(seed + 79, k, 0) * 0.3
    M = np.zeros(nm, _abi.material_dt)
    X_ = np.zeros(nm, _abi.texture_dt)
    M["kind"] = np.where(is_lambert, _abi.MAT_LAMBERTIAN, _abi.MAT_METAL)
    M["texture"] = k
    M["has_fuzz"] = np.where(is_lambert, 0, 1)
    M["fuzz"] = np.where(is_lambert, 0.0, fuzz)
    X_["kind"] = _abi.TEX_SOLID
    X_["rgb"] = np.where(is_lambert[:, None], 0.2 + 0.7 * col, 0.5 + 0.45 * col)
    return SceneDesc(None, _tri_records(a, b, c, mat_ids), M, X_, None, None)


def heightfield_camera(width=1920, height=1080, spp=256, max_depth=50):
    """widecam geometry (src/scenes.rs:70-95)."""
    return widecam(width, height, spp, max_depth)


# ---------------------------------------------------------------------------------------------
# S4: textured terrain + icosphere instances (~5M triangles at the default size)
# ---------------------------------------------------------------------------------------------
def icosphere(subdiv):
    t = (1.0 + math.sqrt(5.0)) / 2.0
    v = [(-1, t, 0), (1, t, 0), (-1, -t, 0), (1, -t, 0), (0, -1, t), (0, 1, t), (0, -1, -t), (0, 1, -t),
         (t, 0, -1), (t, 0, 1), (-t, 0, -1), (-t, 0, 1)]
    f = [(0, 11, 5), (0, 5, 1), (0, 1, 7), (0, 7, 10), (0, 10, 11), (1, 5, 9), (5, 11, 4), (11, 10, 2), (10, 7, 6),
         (7, 1, 8), (3, 9, 4), (3, 4, 2), (3, 2, 6), (3, 6, 8), (3, 8, 9), (4, 9, 5), (2, 4, 11), (6, 2, 10),
         (8, 6, 7), (9, 8, 1)]
    V = np.array(v, dtype=np.float64)
    V /= np.linalg.norm(V, axis=1, keepdims=True)
    F = np.array(f, dtype=np.int64)
    for _ in range(subdiv):
        a, b, c = V[F[:, 0]], V[F[:, 1]], V[F[:, 2]]
        ab, bc, ca = a + b, b + c, c + a
        for m in (ab, bc, ca):
            m /= np.linalg.norm(m, axis=1, keepdims=True)
        n = len(F)
        V = np.concatenate([a, b, c, ab, bc, ca])
        ia, ib, ic = np.arange(n), n + np.arange(n), 2 * n + np.arange(n)
        iab, ibc, ica = 3 * n + np.arange(n), 4 * n + np.arange(n), 5 * n + np.arange(n)
        F = np.concatenate([np.stack([ia, iab, ica], 1), np.stack([ib, ibc, iab], 1),
                            np.stack([ic, ica, ibc], 1), np.stack([iab, ibc, ica], 1)])
    return V[F[:, 0]], V[F[:, 1]], V[F[:, 2]]      # outward-facing (counter-clockwise from outside)


def _lattice_noise(seed, size, cell):
    """Value noise of period-free lattice spacing `cell` sampled on a size x size pixel grid:
    the lattice is hashed once and bilinearly interpolated (smoothstep fade)."""
    n = size // cell + 2
    iy, ix = np.mgrid[0:n, 0:n]
    lat = hash_u01(seed, ix, iy)
    t = (np.arange(size) % cell) / float(cell)
    f = t * t * (3.0 - 2.0 * t)
    i0 = np.arange(size) // cell
    a = lat[i0][:, i0] * (1 - f)[None, :] + lat[i0][:, i0 + 1] * f[None, :]
    b = lat[i0 + 1][:, i0] * (1 - f)[None, :] + lat[i0 + 1][:, i0 + 1] * f[None, :]
    return a * (1 - f)[:, None] + b * f[:, None]


def _noise_texture(seed, size):
    y, x = np.mgrid[0:size, 0:size]
    h = _mix(_mix(np.asarray(seed, dtype=U64)) ^ (x.astype(U64) * U64(0xD6E8FEB86659FD93) + y.astype(U64) * U64(0x9E3779B97F4A7C15)))
    img = np.zeros((size, size, 4), dtype=np.uint8)
    for ch in range(3):
        grain = ((h >> U64(16 * ch)) & U64(0xFFFF)).astype(np.float64) / 65536.0
        n = 0.5 * _lattice_noise(seed + ch, size, 64) + 0.3 * _lattice_noise(seed + 10 + ch, size, 8) + 0.2 * grain
        img[..., ch] = np.clip(np.rint(255.0 * (0.25 + 0.7 * n)), 0, 255).astype(np.uint8)
    return img


def textured_scene(terrain_quads=762, instances=3000, subdiv=3, n_textures=8, tex_size=2048, seed=4, extent=50.0):
    """S4 of SURVEY.md §8d: S3-style terrain + icosphere instances; image-textured Metal with fuzz
    U[0,0.5) (mirrors Material::from_gltf -> Metal(image, Some(roughness)), src/material.rs:25-28);
    10 % of the instances are Dielectric::new_frosted(1.5, 0.05) (src/main.rs:35)."""
    terrain = heightfield_scene(terrain_quads, extent, 5.0, seed, 16)
    X, Y, Z = heightfield_vertices(terrain_quads, extent, 5.0, seed)
    ia, ib, ic = icosphere(subdiv)
    nt = len(ia)
    k = np.arange(instances)
    cx = -extent * 0.9 + 1.8 * extent * hash_u01(seed + 1, k, 0)
    cy = -extent * 0.9 + 1.8 * extent * hash_u01(seed + 1, k, 1)
    rad = 0.3 + 0.9 * hash_u01(seed + 1, k, 2)
    gi = np.clip(np.rint((cx + extent) / (2 * extent) * terrain_quads).astype(int), 0, terrain_quads)
    gj = np.clip(np.rint((cy + extent) / (2 * extent) * terrain_quads).astype(int), 0, terrain_quads)
    cz = Z[gj, gi] + rad * (0.6 + 2.0 * hash_u01(seed + 1, k, 3))
    ctr = np.stack([cx, cy, cz], axis=-1)

    def inst(v):
        return (v[None, :, :] * rad[:, None, None] + ctr[:, None, :]).reshape(-1, 3)

    a, b, c = inst(ia), inst(ib), inst(ic)
    # spherical uv per vertex of the unit icosphere, shared by all instances
    def suv(v):
        return np.stack([np.arctan2(v[:, 1], v[:, 0]) / (2 * math.pi) + 0.5, np.arccos(np.clip(v[:, 2], -1, 1)) / math.pi], -1)

    uv = np.stack([suv(ia), suv(ib), suv(ic)], axis=1)              # (nt, 3, 2)
    frosted = hash_u01(seed + 2, k, 0) < 0.10
    tex_of = (hash_u01(seed + 2, k, 1) * n_textures).astype(np.int64) % n_textures
    fuzz = hash_u01(seed + 2, k, 2) * 0.5

    nm0, nx0 = len(terrain.materials), len(terrain.textures)
    pool = bytearray()
    X_img = np.zeros(n_textures, _abi.texture_dt)
    for t in range(n_textures):
        img = _noise_texture(seed * 100 + t, tex_size)
        X_img[t]["kind"], X_img[t]["width"], X_img[t]["height"] = _abi.TEX_IMAGE, tex_size, tex_size
        X_img[t]["texel_format"], X_img[t]["texel_offset"] = _abi.TEXEL_RGBA8, len(pool)
        pool.extend(img.tobytes())
    M_inst = np.zeros(instances, _abi.material_dt)
    M_inst["kind"] = np.where(frosted, _abi.MAT_DIELECTRIC, _abi.MAT_METAL)
    M_inst["texture"] = nx0 + tex_of
    M_inst["has_fuzz"] = 1
    M_inst["fuzz"] = np.where(frosted, 0.05, fuzz)
    M_inst["ior"] = np.where(frosted, 1.5, 0.0)
    T_inst = _tri_records(a, b, c, np.repeat(nm0 + k, nt).astype(np.uint32))
    T_inst["uv"] = np.tile(uv, (instances, 1, 1))
    return SceneDesc(None, np.concatenate([terrain.triangles, T_inst]),
                     np.concatenate([terrain.materials, M_inst]), np.concatenate([terrain.textures, X_img]),
                     np.frombuffer(bytes(pool), dtype=np.uint8).copy(), None)


# ---------------------------------------------------------------------------------------------
# small mixed scene for tests: spheres + triangles + every material / texture kind
# ---------------------------------------------------------------------------------------------
def mixed_test_scene(seed=11, n_spheres=40, quads=12):
    rng = _Stream(seed)
    img = ImageTexture(procedural_image(64, 32))
    chk = CheckerTexture(0.31, SolidColor((0.0, 0.0, 0.0)), CheckerTexture(0.05, SolidColor((0.9, 0.2, 0.2)), img))
    mats = [Lambertian(chk), Lambertian(img), Metal(img, 0.2), Metal.new_solid((0.8, 0.8, 0.9), None),
            Metal.new_solid((0.7, 0.6, 0.5), 0.0), Dielectric(1.5), Dielectric.new_frosted(1.5, 0.05),
            Lambertian.new_rgb_solid(0.95, 0.70, 0.85), Dielectric.new_inside_other(1.0, 1.5)]
    shapes = generate_ground_plane(60.0, 60.0, -0.2, Lambertian(CheckerTexture(3.0, SolidColor((0.1, 0.1, 0.1)),
                                                                             SolidColor((0.95, 0.95, 0.95)))), True)
    for i in range(n_spheres):
        c = (rng.range(-6, 6), rng.range(-6, 6), rng.range(0.0, 3.0))
        r = rng.range(0.15, 0.9)
        m = mats[i % len(mats)]
        if i % 3 == 0:
            shapes.append(Sphere.new_facing(c, r, m, (rng.range(-1, 1), rng.range(-1, 1), rng.range(-1, 1))))
        else:
            shapes.append(Sphere(c, r, m))
    desc = SceneDesc.from_shapes(shapes)
    hf = heightfield_scene(quads, 7.0, 1.0, seed, 4)
    hf.triangles["a"][:, 2] += 0.3
    hf.triangles["b"][:, 2] += 0.3
    hf.triangles["c"][:, 2] += 0.3
    nm, nx = len(desc.materials), len(desc.textures)
    hm, hx = hf.materials.copy(), hf.textures
    hm["texture"] += nx
    ht = hf.triangles.copy()
    ht["material"] += nm
    return SceneDesc(desc.spheres, np.concatenate([desc.triangles, ht]), np.concatenate([desc.materials, hm]),
                     np.concatenate([desc.textures, hx]), desc.texel_pool, None)
