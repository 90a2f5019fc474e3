"""ctypes wrapper of oracle/liboracle.so — the CPU checker (test infrastructure only)."""
import ctypes as C
import os
import subprocess

import numpy as np

from rt_amd import _abi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
LIB_PATH = os.path.join(ORACLE_DIR, "liboracle.so")

BRUTE, BVH_NOPRUNE, BVH_PRUNE = 0, 1, 2
_lib = None


def build():
    subprocess.run(["make", "-s", "-C", ORACLE_DIR], check=True)


def lib():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        build()
    L = C.CDLL(LIB_PATH)
    vp, sz, u32, u64, i32, dbl = C.c_void_p, C.c_size_t, C.c_uint32, C.c_uint64, C.c_int, C.c_double
    L.orc_scene_create.argtypes = [vp, sz, vp, sz, vp, sz, vp, sz, vp, sz, vp, C.POINTER(vp)]
    L.orc_scene_destroy.argtypes = [vp]
    L.orc_scene_destroy.restype = None
    L.orc_trace_closest.argtypes = [vp, vp, sz, dbl, dbl, i32, vp, vp]
    L.orc_trace_primary.argtypes = [vp, vp, u32, i32, u64, i32, vp, vp]
    L.orc_render.argtypes = [vp, vp, vp, i32, i32, u32, u32, vp, vp]
    L.orc_radiance_sample.argtypes = [vp, vp, u32, u32, u32, u64, i32, vp, vp]
    L.orc_halton_recurrence.argtypes = [u64, u64]
    L.orc_halton_recurrence.restype = dbl
    L.orc_halton_closed.argtypes = [u32, u32]
    L.orc_halton_closed.restype = dbl
    L.orc_philox4x32_10.argtypes = [vp, vp, vp]
    L.orc_philox4x32_10.restype = None
    L.orc_uniform.argtypes = [u64, u32, u32, u32, u32]
    L.orc_uniform.restype = dbl
    L.orc_ray_new.argtypes = [vp, vp, vp, vp]
    L.orc_ray_new.restype = None
    L.orc_sphere_hit.argtypes = [vp, vp, dbl, dbl, vp]
    L.orc_triangle_hit.argtypes = [vp, vp, dbl, dbl, vp]
    L.orc_reflect.argtypes = [vp, vp, vp]
    L.orc_reflect.restype = None
    L.orc_refract.argtypes = [vp, vp, dbl, vp]
    L.orc_refract.restype = None
    L.orc_reflectance.argtypes = [dbl, dbl]
    L.orc_reflectance.restype = dbl
    L.orc_texture_value.argtypes = [vp, u32, dbl, dbl, vp, vp]
    L.orc_texture_value.restype = None
    L.orc_uncharted2.argtypes = [vp, vp]
    L.orc_uncharted2.restype = None
    L.orc_sky_color.argtypes = [vp, vp, vp]
    L.orc_sky_color.restype = None
    L.orc_sky_radiance.argtypes = [vp, C.c_float, C.c_float, i32]
    L.orc_sky_radiance.restype = C.c_float
    L.orc_scatter.argtypes = [vp, vp, vp, u64, u32, u32, u32, vp, vp]
    L.orc_sphere_uv.argtypes = [vp, vp, vp]
    L.orc_sphere_uv.restype = None
    L.orc_gamma8.argtypes = [vp, sz, vp]
    L.orc_gamma8.restype = None
    L.orc_last_render_segments.restype = u64
    _lib = L
    return L


def f3(v):
    return np.ascontiguousarray(v, dtype=np.float64).reshape(3)


class OracleScene:
    def __init__(self, desc):
        self.desc = desc
        self._h = C.c_void_p()
        rc = lib().orc_scene_create(
            _abi.ptr(desc.spheres), len(desc.spheres), _abi.ptr(desc.triangles), len(desc.triangles),
            _abi.ptr(desc.materials), len(desc.materials), _abi.ptr(desc.textures), len(desc.textures),
            _abi.ptr(desc.texel_pool), desc.texel_pool.nbytes, _abi.ptr(desc.sky), C.byref(self._h))
        if rc != 0:
            raise RuntimeError(f"orc_scene_create failed: {rc}")

    def hit(self, rays, t_min=0.001, t_max=1.7976931348623157e308, mode=BRUTE, with_isect=False):
        r = np.asarray(rays)
        if r.dtype != _abi.ray_dt:
            r = np.ascontiguousarray(r, dtype=np.float64).reshape(-1, 6).view(_abi.ray_dt).reshape(-1)
        r = np.ascontiguousarray(r)
        out = np.zeros(len(r), _abi.hit_dt)
        iso = np.zeros(len(r), _abi.isect_dt) if with_isect else None
        rc = lib().orc_trace_closest(self._h, _abi.ptr(r), len(r), t_min, t_max, mode, _abi.ptr(out), _abi.ptr(iso))
        assert rc == 0, rc
        return (out, iso) if with_isect else out

    def trace_primary(self, cam, sample_index=0, jitter=True, seed=0, mode=BRUTE, with_rays=False):
        n = cam.width * cam.height
        out = np.zeros(n, _abi.hit_dt)
        rays = np.zeros(n, _abi.ray_dt) if with_rays else None
        rc = lib().orc_trace_primary(self._h, _abi.ptr(cam.rec), sample_index, int(bool(jitter)), seed, mode,
                                     _abi.ptr(out), _abi.ptr(rays))
        assert rc == 0, rc
        return (out, rays) if with_rays else out

    def render(self, cam, spp, seed=0, mode=BVH_PRUNE, nthreads=8, sample_offset=0, sample_stride=1, rows=None,
               with_stats=False):
        o = cam.opts(spp=spp, seed=seed, sample_offset=sample_offset, sample_stride=sample_stride)
        out = np.zeros((cam.height, cam.width, 3), dtype=np.float64)
        st = _abi.one(_abi.stats_dt)
        y0, y1 = (0, cam.height) if rows is None else rows
        rc = lib().orc_render(self._h, _abi.ptr(cam.rec), _abi.ptr(o), mode, nthreads, y0, y1, _abi.ptr(out), _abi.ptr(st))
        assert rc == 0, rc
        return (out, st[0]) if with_stats else out

    def radiance_sample(self, cam, x, y, s, seed=0, mode=BRUTE):
        out = np.zeros(3)
        nseg = C.c_uint32()
        rc = lib().orc_radiance_sample(self._h, _abi.ptr(cam.rec), x, y, s, seed, mode, _abi.ptr(out), C.byref(nseg))
        assert rc == 0, rc
        return out, nseg.value

    def texture_value(self, tex, u, v, p):
        out = np.zeros(3)
        lib().orc_texture_value(self._h, tex, u, v, _abi.ptr(f3(p)), _abi.ptr(out))
        return out

    def sky_color(self, d):
        out = np.zeros(3)
        lib().orc_sky_color(self._h, _abi.ptr(f3(d)), _abi.ptr(out))
        return out

    def close(self):
        if self._h:
            lib().orc_scene_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
