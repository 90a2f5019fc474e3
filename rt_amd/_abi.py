"""numpy / ctypes views of the POD structs declared in include/rt_hip.h.

The numpy dtypes are laid out exactly like the C structs (sizes are static_assert'ed in
rt_amd/csrc/rt_api.hip and checked again in tests/test_abi.py), so a structured array's buffer can
be handed to the C ABI without copying.
"""
import ctypes as C

import numpy as np

RT_MISS = 0xFFFFFFFF
RT_MAX_SAMPLE_INDEX = 1 << 20

RT_OK, RT_ERR_INVALID, RT_ERR_NO_DEVICE, RT_ERR_HIP, RT_ERR_OOM, RT_ERR_LIMIT = 0, -1, -2, -3, -4, -5

MAT_LAMBERTIAN, MAT_METAL, MAT_DIELECTRIC = 0, 1, 2
TEX_SOLID, TEX_CHECKER, TEX_IMAGE = 0, 1, 2
TEXEL_RGBA8, TEXEL_RGBA16 = 0, 1

sphere_dt = np.dtype([("center", "<f8", 3), ("radius", "<f8"), ("face", "<f8", 3),
                      ("material", "<u4"), ("_pad", "<u4")])
triangle_dt = np.dtype([("a", "<f8", 3), ("b", "<f8", 3), ("c", "<f8", 3), ("uv", "<f8", (3, 2)),
                        ("n", "<f8", 3), ("material", "<u4"), ("_pad", "<u4")])
material_dt = np.dtype([("kind", "<u4"), ("texture", "<u4"), ("has_fuzz", "<u4"), ("_pad", "<u4"),
                        ("fuzz", "<f8"), ("ior", "<f8")])
texture_dt = np.dtype([("kind", "<u4"), ("even", "<u4"), ("odd", "<u4"), ("width", "<u4"),
                       ("height", "<u4"), ("texel_format", "<u4"), ("texel_offset", "<u8"),
                       ("rgb", "<f8", 3), ("inv_scale", "<f8")])
sky_dt = np.dtype([("params", "<f4", 27), ("radiances", "<f4", 3), ("sun", "<f8", 3)])
camera_dt = np.dtype([("center", "<f8", 3), ("pixel00", "<f8", 3), ("pixel_du", "<f8", 3),
                      ("pixel_dv", "<f8", 3), ("defocus_u", "<f8", 3), ("defocus_v", "<f8", 3),
                      ("defocus_angle", "<f8"), ("t_min", "<f8"), ("t_max", "<f8"),
                      ("width", "<u4"), ("height", "<u4"), ("max_depth", "<u4"), ("_pad", "<u4")])
opts_dt = np.dtype([("spp", "<u4"), ("sample_offset", "<u4"), ("sample_stride", "<u4"),
                    ("precision", "<u4"), ("seed", "<u8"), ("device", "<i4"),
                    ("max_paths_in_flight", "<u4"), ("collect_stats", "<u4"), ("flags", "<u4")])
OPT_REDUCE_F32 = 1
OPT_PIPELINED = 2
stats_dt = np.dtype([("samples", "<u8"), ("ray_segments", "<u8"), ("nodes_visited", "<u8"),
                     ("prims_tested", "<u8"), ("passes", "<u4"), ("launches", "<u4"),
                     ("seconds_trace", "<f8"), ("seconds_shade", "<f8"), ("seconds_total", "<f8"),
                     ("trace_launches", "<u8"), ("primary_segments", "<u8"), ("seconds_trace_primary", "<f8"),
                     ("nodes_visited_primary", "<u8"), ("prims_tested_primary", "<u8"),
                     ("bounce_segments", "<u8"), ("seconds_trace_bounce", "<f8"), ("bounce_launches", "<u8"),
                     ("tail_segments", "<u8"), ("seconds_tail", "<f8"),
                     ("nodes_visited_bounce", "<u8"), ("prims_tested_bounce", "<u8"),
                     ("wave_node_fetches_primary", "<u8"), ("wave_prim_fetches_primary", "<u8")])
ray_dt = np.dtype([("o", "<f8", 3), ("d", "<f8", 3)])
hit_dt = np.dtype([("t", "<f8"), ("prim", "<u4"), ("_pad", "<u4")])
scene_info_dt = np.dtype([("n_prims", "<u8"), ("n_nodes", "<u8"), ("device_bytes", "<u8"),
                          ("bvh_depth", "<u4"), ("max_leaf_size", "<u4"), ("node_bytes", "<u4"),
                          ("tri_bytes", "<u4"), ("sphere_bytes", "<u4"), ("bvh_width", "<u4"),
                          ("build_seconds", "<f8"), ("upload_seconds", "<f8"), ("builder", "<u4"), ("_pad", "<u4")])
# oracle-only record (oracle/rt_oracle.h)
isect_dt = np.dtype([("point", "<f8", 3), ("normal", "<f8", 3), ("uv", "<f8", 2), ("t", "<f8"),
                     ("prim", "<u4"), ("material", "<u4"), ("front", "<i4"), ("_pad", "<i4")])

# sizes the C side static_asserts
EXPECTED_SIZES = {
    "rt_sphere": (sphere_dt, 64), "rt_triangle": (triangle_dt, 152), "rt_material": (material_dt, 32),
    "rt_texture": (texture_dt, 64), "rt_sky": (sky_dt, 144), "rt_camera": (camera_dt, 184),
    "rt_opts": (opts_dt, 40), "rt_stats": (stats_dt, 176), "rt_ray": (ray_dt, 48), "rt_hit": (hit_dt, 16),
    "rt_scene_info": (scene_info_dt, 72),
}
for _name, (_dt, _size) in EXPECTED_SIZES.items():
    assert _dt.itemsize == _size, (_name, _dt.itemsize, _size)


def ptr(arr):
    """void* of a C-contiguous numpy array (None -> NULL)."""
    if arr is None:
        return None
    assert arr.flags["C_CONTIGUOUS"]
    return arr.ctypes.data_as(C.c_void_p)


def one(dt):
    """A zero-initialised single record of dtype dt, as a 1-element array."""
    return np.zeros(1, dtype=dt)
