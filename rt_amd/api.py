"""Host-side mirror of the reference's scene / camera API for the render hot path.

The reference is a Rust crate whose toolchain is absent from this image, so the host side above the
C ABI is mirrored here with the reference's names and argument meaning:

    Camera.new(...)                       src/camera.rs:169-227
    Camera.render_image(world)            src/camera.rs:327-341
    Camera.render_pixel(world,x,y,n)      src/camera.rs:315-325
    Camera.debug_ray / debug_raycast      src/camera.rs:256-276
    World.build(shapes)                   src/hittable.rs:33-46
    World.hit(ray, t_min, t_max)          src/hittable.rs:135-149
    Sphere.new / new_facing               src/hittable.rs:270-297
    Triangle.new / new_with_uv            src/hittable.rs:165-208
    Lambertian / Metal / Dielectric       src/material.rs:62-148
    SolidColor / CheckerTexture / ImageTexture   src/texture.rs

Everything here is flattening and bookkeeping in f64 with the reference's operation order; all
rendering goes through librt_hip.so (rt_amd/_lib.py).  There is no CPU render path in this package.
"""
import ctypes as C
import math

import numpy as np

from . import _abi
from ._lib import check, lib
from .sky import sky_synth_v1

F64_MAX = 1.7976931348623157e308


# ---------------------------------------------------------------------------------------------
# nalgebra-ordered scalar helpers: dot = (x0*y0 + x1*y1) + x2*y2, normalize = v / norm
# ---------------------------------------------------------------------------------------------
def _dot(a, b):
    return (a[0] * b[0] + a[1] * b[1]) + a[2] * b[2]


def _sub(a, b):
    return (a[0] - b[0], a[1] - b[1], a[2] - b[2])


def _add(a, b):
    return (a[0] + b[0], a[1] + b[1], a[2] + b[2])


def _scale(a, s):
    return (a[0] * s, a[1] * s, a[2] * s)


def _fdiv(x, s):
    """IEEE division like Rust's f64 (0/0 = NaN, x/0 = +-inf): a zero-area triangle gets a NaN normal, not an exception."""
    try:
        return x / s
    except ZeroDivisionError:
        if x == 0.0 or x != x:
            return math.nan
        return math.copysign(math.inf, x) * math.copysign(1.0, s)


def _div(a, s):
    return (_fdiv(a[0], s), _fdiv(a[1], s), _fdiv(a[2], s))


def _cross(a, b):
    return (a[1] * b[2] - a[2] * b[1], a[2] * b[0] - a[0] * b[2], a[0] * b[1] - a[1] * b[0])


def _normalize(a):
    return _div(a, math.sqrt(_dot(a, a)))


def _f3(v):
    return (float(v[0]), float(v[1]), float(v[2]))


def triangle_normals(a, b, c):
    """Triangle::new's stored normal for arrays of vertices (src/hittable.rs:169-178):
    normalize(cross(normalize(b-a), normalize(c-a))), nalgebra operation order, f64."""
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    c = np.asarray(c, dtype=np.float64)

    def nrm(v):
        n = np.sqrt((v[..., 0] * v[..., 0] + v[..., 1] * v[..., 1]) + v[..., 2] * v[..., 2])
        return v / n[..., None]

    ab = nrm(b - a)
    ac = nrm(c - a)
    cr = np.stack([ab[..., 1] * ac[..., 2] - ab[..., 2] * ac[..., 1],
                   ab[..., 2] * ac[..., 0] - ab[..., 0] * ac[..., 2],
                   ab[..., 0] * ac[..., 1] - ab[..., 1] * ac[..., 0]], axis=-1)
    return nrm(cr)


# ---------------------------------------------------------------------------------------------
# Textures (src/texture.rs) and materials (src/material.rs)
# ---------------------------------------------------------------------------------------------
class SolidColor:
    def __init__(self, color):
        self.color = _f3(color)

    @staticmethod
    def new_rgb(r, g, b):
        return SolidColor((r, g, b))


class CheckerTexture:
    def __init__(self, scale, even_texture, odd_texture):
        self.scale_inverted = 1.0 / float(scale)       # src/texture.rs:54
        self.even_texture = even_texture
        self.odd_texture = odd_texture


class ImageTexture:
    """image: (H, W, 3|4) uint8 (channel/255.0, src/camera.rs:68-70) or uint16 (channel/65535.0, :92-93)."""

    def __init__(self, image):
        img = np.asarray(image)
        if img.ndim != 3 or img.shape[2] not in (3, 4) or img.dtype not in (np.uint8, np.uint16):
            raise ValueError("ImageTexture expects (H, W, 3|4) uint8 or uint16")
        if img.shape[0] < 1 or img.shape[1] < 1:
            raise ValueError("empty image")
        self.image = img


class Lambertian:
    def __init__(self, texture):
        self.texture = texture

    @staticmethod
    def new_rgb_solid(r, g, b):
        return Lambertian(SolidColor.new_rgb(r, g, b))


class Metal:
    def __init__(self, texture, fuzz=None):
        self.texture = texture
        self.fuzz = None if fuzz is None else float(fuzz)

    @staticmethod
    def new_solid(color, fuzz=None):
        return Metal(SolidColor(color), fuzz)


class Dielectric:
    def __init__(self, refractive_index, fuzz=None):
        self.refractive_index = float(refractive_index)
        self.fuzz = None if fuzz is None else float(fuzz)

    @staticmethod
    def new_frosted(refractive_index, fuzz):
        return Dielectric(refractive_index, fuzz)

    @staticmethod
    def new_inside_other(material_index, container_index):
        return Dielectric(float(material_index) / float(container_index))


# ---------------------------------------------------------------------------------------------
# Shapes (src/hittable.rs)
# ---------------------------------------------------------------------------------------------
class Sphere:
    def __init__(self, center, radius, material, front_direction=(1.0, 0.0, 0.0)):
        self.center = _f3(center)
        self.radius = max(float(radius), 0.0)          # :275
        self.material = material
        self.front_direction = _f3(front_direction)    # :278

    @staticmethod
    def new_facing(center, radius, material, front_face):
        return Sphere(center, radius, material, front_face)


class Triangle:
    def __init__(self, a, b, c, material, uv_a=(0.0, 0.0), uv_b=(1.0, 0.0), uv_c=(0.5, 1.0)):
        self.a, self.b, self.c = _f3(a), _f3(b), _f3(c)
        self.uv_a, self.uv_b, self.uv_c = tuple(map(float, uv_a)), tuple(map(float, uv_b)), tuple(map(float, uv_c))
        ab = _normalize(_sub(self.b, self.a))
        ac = _normalize(_sub(self.c, self.a))
        self.normal = _normalize(_cross(ab, ac))       # :169-178
        self.material = material

    @staticmethod
    def new_with_uv(a, b, c, uv_a, uv_b, uv_c, material):
        return Triangle(a, b, c, material, uv_a, uv_b, uv_c)

    @staticmethod
    def new_opposite_normal(a, b, c, material):
        return Triangle(c, b, a, material)             # :210-212


def generate_ground_plane(width, length, z, material, top_is_up=True):
    """scenes::generate_ground_plane (src/scenes.rs:310-342)."""
    hw, hl = width / 2.0, length / 2.0
    a, b, c, d = (-hw, -hl, z), (hw, -hl, z), (hw, hl, z), (-hw, hl, z)
    if top_is_up:
        return [Triangle(a, b, c, material), Triangle(a, c, d, material)]
    return [Triangle.new_opposite_normal(a, b, c, material), Triangle.new_opposite_normal(a, c, d, material)]


# ---------------------------------------------------------------------------------------------
# Flattened scene description: exactly the arrays rt_scene_create takes
# ---------------------------------------------------------------------------------------------
class SceneDesc:
    def __init__(self, spheres=None, triangles=None, materials=None, textures=None, texel_pool=None, sky=None):
        self.spheres = np.zeros(0, _abi.sphere_dt) if spheres is None else np.ascontiguousarray(spheres, _abi.sphere_dt)
        self.triangles = np.zeros(0, _abi.triangle_dt) if triangles is None else np.ascontiguousarray(triangles, _abi.triangle_dt)
        self.materials = np.zeros(0, _abi.material_dt) if materials is None else np.ascontiguousarray(materials, _abi.material_dt)
        self.textures = np.zeros(0, _abi.texture_dt) if textures is None else np.ascontiguousarray(textures, _abi.texture_dt)
        self.texel_pool = np.zeros(0, np.uint8) if texel_pool is None else np.ascontiguousarray(texel_pool, np.uint8)
        self.sky = sky_synth_v1() if sky is None else sky
        # prim id (the [spheres..., triangles...] order of the ABI) -> position in the caller's shapes list, and
        # the index inside that entry when it is a TriangleSoup; None when the arrays were given directly
        self.prim_shape = None
        self.prim_sub = None

    def shape_of(self, prim):
        """(index in the shapes list handed to from_shapes, index inside that entry) of primitive id(s) `prim`
        — the reference's `Vec<Shape>` position for mixed sphere / triangle lists."""
        if self.prim_shape is None:
            raise ValueError("this SceneDesc was not built by from_shapes")
        p = np.asarray(prim, dtype=np.int64)
        return self.prim_shape[p], self.prim_sub[p]

    @property
    def n_prims(self):
        return len(self.spheres) + len(self.triangles)

    @staticmethod
    def from_shapes(shapes, sky=None):
        """Flatten a list of Sphere/Triangle objects the way the Rust shim would flatten
        `Vec<Shape>`: materials de-duplicated by identity (shared `Arc<Material>`), textures
        flattened to a node table with children before parents, images packed as RGBA."""
        mats, mat_ids = [], {}
        texs, tex_ids = [], {}
        pool = bytearray()

        def tex_id(t):
            if id(t) in tex_ids:
                return tex_ids[id(t)]
            rec = np.zeros(1, _abi.texture_dt)[0]
            if isinstance(t, SolidColor):
                rec["kind"] = _abi.TEX_SOLID
                rec["rgb"] = t.color
            elif isinstance(t, CheckerTexture):
                e, o = tex_id(t.even_texture), tex_id(t.odd_texture)
                rec["kind"] = _abi.TEX_CHECKER
                rec["even"], rec["odd"] = e, o
                rec["inv_scale"] = t.scale_inverted
            elif isinstance(t, ImageTexture):
                img = t.image
                h, w = img.shape[:2]
                rgba = np.zeros((h, w, 4), dtype=img.dtype)
                rgba[..., :img.shape[2]] = img
                while len(pool) % 8:
                    pool.append(0)
                rec["kind"] = _abi.TEX_IMAGE
                rec["width"], rec["height"] = w, h
                rec["texel_format"] = _abi.TEXEL_RGBA16 if img.dtype == np.uint16 else _abi.TEXEL_RGBA8
                rec["texel_offset"] = len(pool)
                pool.extend(rgba.tobytes())
            else:
                raise TypeError(f"not a texture: {t!r}")
            texs.append(rec)
            tex_ids[id(t)] = len(texs) - 1
            return tex_ids[id(t)]

        def mat_id(m):
            if id(m) in mat_ids:
                return mat_ids[id(m)]
            rec = np.zeros(1, _abi.material_dt)[0]
            if isinstance(m, Lambertian):
                rec["kind"], rec["texture"] = _abi.MAT_LAMBERTIAN, tex_id(m.texture)
            elif isinstance(m, Metal):
                rec["kind"], rec["texture"] = _abi.MAT_METAL, tex_id(m.texture)
                rec["has_fuzz"] = 0 if m.fuzz is None else 1
                rec["fuzz"] = 0.0 if m.fuzz is None else m.fuzz
            elif isinstance(m, Dielectric):
                rec["kind"] = _abi.MAT_DIELECTRIC
                rec["ior"] = m.refractive_index
                rec["has_fuzz"] = 0 if m.fuzz is None else 1
                rec["fuzz"] = 0.0 if m.fuzz is None else m.fuzz
            else:
                raise TypeError(f"not a material: {m!r}")
            mats.append(rec)
            mat_ids[id(m)] = len(mats) - 1
            return mat_ids[id(m)]

        from .importers import TriangleSoup             # arrays of triangles (importers): flattened without a loop
        sph = [s for s in shapes if isinstance(s, Sphere)]
        tri = [t for t in shapes if isinstance(t, (Triangle, TriangleSoup))]
        if len(sph) + len(tri) != len(shapes):
            raise TypeError("shapes must be Sphere, Triangle or TriangleSoup")
        # the ABI orders primitives [spheres..., triangles...]; remember where each one sat in `shapes`
        sph_pos = [i for i, s in enumerate(shapes) if isinstance(s, Sphere)]
        tri_pos, tri_sub = [], []
        for i, t in enumerate(shapes):
            if isinstance(t, Triangle):
                tri_pos.append(np.array([i], dtype=np.int64))
                tri_sub.append(np.zeros(1, dtype=np.int64))
            elif isinstance(t, TriangleSoup):
                tri_pos.append(np.full(len(t), i, dtype=np.int64))
                tri_sub.append(np.arange(len(t), dtype=np.int64))
        S = np.zeros(len(sph), _abi.sphere_dt)
        for i, s in enumerate(sph):
            S[i]["center"], S[i]["radius"], S[i]["face"], S[i]["material"] = s.center, s.radius, s.front_direction, mat_id(s.material)
        parts = []
        single = []

        def flush_single():
            if single:
                T1 = np.zeros(len(single), _abi.triangle_dt)
                for i, t in enumerate(single):
                    T1[i]["a"], T1[i]["b"], T1[i]["c"] = t.a, t.b, t.c
                    T1[i]["uv"] = (t.uv_a, t.uv_b, t.uv_c)
                    T1[i]["n"], T1[i]["material"] = t.normal, mat_id(t.material)
                parts.append(T1)
                single.clear()

        for t in tri:
            if isinstance(t, Triangle):
                single.append(t)
                continue
            flush_single()
            Tn = np.zeros(len(t), _abi.triangle_dt)
            Tn["a"], Tn["b"], Tn["c"], Tn["uv"], Tn["n"] = t.a, t.b, t.c, t.uv, t.normal
            Tn["material"] = mat_id(t.material)
            parts.append(Tn)
        flush_single()
        T = np.concatenate(parts) if parts else np.zeros(0, _abi.triangle_dt)
        M = np.array(mats, dtype=_abi.material_dt) if mats else np.zeros(0, _abi.material_dt)
        X = np.array(texs, dtype=_abi.texture_dt) if texs else np.zeros(0, _abi.texture_dt)
        desc = SceneDesc(S, T, M, X, np.frombuffer(bytes(pool), dtype=np.uint8).copy(), sky)
        desc.prim_shape = np.concatenate([np.array(sph_pos, dtype=np.int64)] + tri_pos)
        desc.prim_sub = np.concatenate([np.zeros(len(sph_pos), dtype=np.int64)] + tri_sub)
        return desc


# ---------------------------------------------------------------------------------------------
# World (src/hittable.rs:24-46, :135-149)
# ---------------------------------------------------------------------------------------------
class World:
    def __init__(self, handle, desc, device):
        self._h = handle
        self.desc = desc
        self.device = device

    @staticmethod
    def build(shapes, device=0, sky=None):
        """World::build: flatten, build the BVH, upload to HBM of `device`."""
        desc = shapes if isinstance(shapes, SceneDesc) else SceneDesc.from_shapes(shapes, sky)
        h = C.c_void_p()
        check(lib().rt_scene_create(
            _abi.ptr(desc.spheres), len(desc.spheres), _abi.ptr(desc.triangles), len(desc.triangles),
            _abi.ptr(desc.materials), len(desc.materials), _abi.ptr(desc.textures), len(desc.textures),
            _abi.ptr(desc.texel_pool), desc.texel_pool.nbytes, _abi.ptr(desc.sky), int(device), C.byref(h)))
        return World(h, desc, device)

    def info(self):
        rec = _abi.one(_abi.scene_info_dt)
        check(lib().rt_scene_get_info(self._h, _abi.ptr(rec)))
        return {k: rec[k][0].item() for k in rec.dtype.names if k != "_pad"}

    def hit(self, rays, t_min=0.001, t_max=F64_MAX, with_stats=False):
        """World::hit for a batch: rays is a ray_dt array or (n, 6) floats (o, d); returns a hit_dt array."""
        r = np.asarray(rays)
        if r.dtype != _abi.ray_dt:
            r = np.ascontiguousarray(r, dtype=np.float64).reshape(-1, 6).view(_abi.ray_dt).reshape(-1)
        r = np.ascontiguousarray(r)
        out = np.zeros(len(r), _abi.hit_dt)
        # (a stats pointer selects the instrumented traversal kernels: only when asked for)
        st = _abi.one(_abi.stats_dt) if with_stats else None
        check(lib().rt_trace_closest(self._h, _abi.ptr(r), len(r), t_min, t_max, _abi.ptr(out), _abi.ptr(st)))
        return (out, st[0]) if with_stats else out

    def sync(self):
        """rt_scene_sync: waits for the frames queued with opts(pipelined=True); returns the stats no call has reported yet."""
        st = _abi.one(_abi.stats_dt)
        check(lib().rt_scene_sync(self._h, _abi.ptr(st)))
        return st[0]

    def close(self):
        if self._h:
            lib().rt_scene_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


# ---------------------------------------------------------------------------------------------
# Camera (src/camera.rs)
# ---------------------------------------------------------------------------------------------
class Camera:
    def __init__(self):
        self.rec = _abi.one(_abi.camera_dt)
        self.samples_per_pixel = 1

    @staticmethod
    def new(center, lookat, up, focus_distance, defocus_angle, image_width, image_height,
            samples_per_pixel, max_depth, vertical_fov, t_range=(0.0, F64_MAX)):
        """Camera::new (src/camera.rs:169-227), same positional arguments, f64, same operation order."""
        center, lookat, up = _f3(center), _f3(lookat), _f3(up)
        focus_distance, defocus_angle, vertical_fov = float(focus_distance), float(defocus_angle), float(vertical_fov)
        rad = math.pi / 180.0                                   # f64::to_radians = self * (PI / 180.0)
        w = _normalize(_sub(center, lookat))
        u = _normalize(_cross(up, w))
        v = _cross(w, u)
        h = math.tan((vertical_fov * rad) / 2.0)
        viewport_height = 2.0 * h * focus_distance
        aspect_ratio = float(image_width) / float(image_height)
        viewport_width = viewport_height * aspect_ratio
        viewport_u = _scale(u, viewport_width)
        viewport_v = _scale((-v[0], -v[1], -v[2]), viewport_height)
        pixel_du = _div(viewport_u, float(image_width))
        pixel_dv = _div(viewport_v, float(image_height))
        vp_upper_left = _sub(_sub(_sub(center, _scale(w, focus_distance)), _div(viewport_u, 2.0)), _div(viewport_v, 2.0))
        pixel00 = _add(vp_upper_left, _div(_add(pixel_du, pixel_dv), 2.0))
        defocus_radius = focus_distance * math.tan((defocus_angle / 2.0) * rad)
        cam = Camera()
        r = cam.rec[0]
        r["center"], r["pixel00"], r["pixel_du"], r["pixel_dv"] = center, pixel00, pixel_du, pixel_dv
        r["defocus_u"], r["defocus_v"] = _scale(u, defocus_radius), _scale(v, defocus_radius)
        r["defocus_angle"] = defocus_angle
        r["t_min"] = 0.001                                       # raycast's range start (:297)
        r["t_max"] = float(t_range[1])
        r["width"], r["height"], r["max_depth"] = int(image_width), int(image_height), int(max_depth)
        cam.samples_per_pixel = int(samples_per_pixel)
        return cam

    width = property(lambda self: int(self.rec["width"][0]))
    height = property(lambda self: int(self.rec["height"][0]))
    max_depth = property(lambda self: int(self.rec["max_depth"][0]))

    def opts(self, spp=None, sample_offset=0, sample_stride=1, seed=0, device=0, max_paths_in_flight=0,
             collect_stats=False, reduce_f32=False, pipelined=False):
        o = _abi.one(_abi.opts_dt)
        # reduce_f32: rt_render_multi only, f32 exchange of the per-device sums; pipelined: rt_render_accumulate_device
        # only, the frame is queued and not waited for (World.sync / rt_scene_sync waits)
        o["flags"] = (_abi.OPT_REDUCE_F32 if reduce_f32 else 0) | (_abi.OPT_PIPELINED if pipelined else 0)
        o["spp"] = self.samples_per_pixel if spp is None else spp
        o["sample_offset"], o["sample_stride"] = sample_offset, sample_stride
        o["precision"], o["seed"], o["device"] = 64, seed, device
        o["max_paths_in_flight"], o["collect_stats"] = max_paths_in_flight, int(collect_stats)
        return o

    def render_image(self, world, spp=None, seed=0, with_stats=False, **kw):
        """Camera::render_image: (H, W, 3) f64 linear mean radiance."""
        o = self.opts(spp=spp, seed=seed, device=world.device, **kw)
        out = np.zeros((self.height, self.width, 3), dtype=np.float64)
        st = _abi.one(_abi.stats_dt)
        check(lib().rt_render(world._h, _abi.ptr(self.rec), _abi.ptr(o), _abi.ptr(out), _abi.ptr(st)))
        return (out, st[0]) if with_stats else out

    def render_image_multi(self, worlds, spp=None, seed=0, with_stats=False, **kw):
        """Camera::render_image on several GPUs of this process: worlds[g] is the same scene built on device g
        (World.build(desc, device=g)); rt_render_multi partitions the samples by index and sum-reduces over RCCL."""
        o = self.opts(spp=spp, seed=seed, device=worlds[0].device, **kw)
        out = np.zeros((self.height, self.width, 3), dtype=np.float64)
        st = np.zeros(len(worlds), _abi.stats_dt)
        handles = (C.c_void_p * len(worlds))(*[w._h for w in worlds])
        check(lib().rt_render_multi(handles, len(worlds), _abi.ptr(self.rec), _abi.ptr(o), _abi.ptr(out), _abi.ptr(st)))
        return (out, st) if with_stats else out

    def render_pass(self, world, n_samples, total_samples, seed=0, quantise=True, with_mean=False, with_stats=False):
        """One sweep of window::render_thread (src/window.rs:256-311) through rt_render_pass: the scene keeps the
        frame on the device; returns the (H, W, 4) uint8 linear RGBA frame (and the float frame, and stats)."""
        rgba = np.zeros((self.height, self.width, 4), dtype=np.uint8)
        mean = np.zeros((self.height, self.width, 3), dtype=np.float64) if with_mean else None
        st = _abi.one(_abi.stats_dt)
        check(lib().rt_render_pass(world._h, _abi.ptr(self.rec), int(n_samples), int(total_samples), int(seed),
                                   1 if quantise else 0, _abi.ptr(rgba), _abi.ptr(mean), _abi.ptr(st)))
        out = (rgba,) + ((mean,) if with_mean else ()) + ((st[0],) if with_stats else ())
        return out[0] if len(out) == 1 else out

    def render_pixel(self, world, x, y, num_samples, seed=0):
        """Camera::render_pixel(world, x, y, n): served from a 1-pixel window of the same kernels
        (the camera record is shifted so that pixel (0,0) is (x,y); RNG keys use the real pixel
        index only in full-frame renders, so this matches render_image only in distribution)."""
        sub = Camera()
        sub.rec[:] = self.rec
        r = sub.rec[0]
        p00 = np.array(r["pixel00"]) + np.array(r["pixel_du"]) * float(x) + np.array(r["pixel_dv"]) * float(y)
        r["pixel00"], r["width"], r["height"] = p00, 1, 1
        return sub.render_image(world, spp=num_samples, seed=seed)[0, 0]

    def trace_primary(self, world, sample_index=0, jitter=True, seed=0, with_rays=False):
        """get_ray(x, y, i) + World::hit for every pixel (jitter=False: debug_ray at pixel centres)."""
        n = self.width * self.height
        out = np.zeros(n, _abi.hit_dt)
        rays = np.zeros(n, _abi.ray_dt) if with_rays else None
        check(lib().rt_trace_primary(world._h, _abi.ptr(self.rec), int(sample_index), int(bool(jitter)), int(seed),
                                     _abi.ptr(out), _abi.ptr(rays)))
        return (out, rays) if with_rays else out

    def debug_ray(self, x, y):
        """Camera::debug_ray (src/camera.rs:256-260): un-normalised (o, d) through pixel coords (x, y)."""
        r = self.rec[0]
        ps = np.array(r["pixel00"]) + np.array(r["pixel_du"]) * float(x) + np.array(r["pixel_dv"]) * float(y)
        c = np.array(r["center"])
        return np.concatenate([c, ps - c])

    def debug_raycast(self, world, ray):
        """Camera::debug_raycast (src/camera.rs:262-276), closest-hit part: (prim, t) or None."""
        h = world.hit(np.asarray(ray, dtype=np.float64).reshape(1, 6), 0.001, float(self.rec["t_max"][0]))[0]
        return None if h["prim"] == _abi.RT_MISS else (int(h["prim"]), float(h["t"]))


def encode_gamma8(rgb_mean):
    """Vec3Ext::as_rgb_gamma over an image (src/vec3.rs:71-97), clamping instead of panicking."""
    a = np.ascontiguousarray(rgb_mean, dtype=np.float64)
    out = np.zeros(a.shape, dtype=np.uint8)
    check(lib().rt_encode_gamma8(_abi.ptr(a), a.size, _abi.ptr(out)))
    return out


def write_ppm(path, rgb8):
    """Camera::write_image (src/camera.rs:343-364): ASCII P3 with the reference's header text."""
    h, w = rgb8.shape[:2]
    with open(path, "w") as f:
        f.write(f"P3\n{w} {h} # width, height\n255 # max color value\n")
        for y in range(h):
            f.write(" ".join(f"{r} {g} {b}" for r, g, b in rgb8[y]) + "\n")


def write_png(path, rgb8):
    """The same 8-bit frame as a PNG (the reference only writes ASCII PPM): 8-bit RGB, filter 0,
    one zlib stream; stdlib only."""
    import struct
    import zlib
    img = np.ascontiguousarray(rgb8, dtype=np.uint8)
    if img.ndim != 3 or img.shape[2] != 3:
        raise ValueError("write_png expects (H, W, 3) uint8")
    h, w = img.shape[:2]
    raw = np.concatenate([np.zeros((h, 1), np.uint8), img.reshape(h, w * 3)], axis=1).tobytes()

    def chunk(tag, data):
        return struct.pack(">I", len(data)) + tag + data + struct.pack(">I", zlib.crc32(tag + data) & 0xFFFFFFFF)

    with open(path, "wb") as f:
        f.write(b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, 8, 2, 0, 0, 0)) +
                chunk(b"IDAT", zlib.compress(raw, 6)) + chunk(b"IEND", b""))
