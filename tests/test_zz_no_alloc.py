"""rt_trace_closest_noalloc (the entry point for callers that cannot allocate): through the C ABI, against
rt_trace_closest on the same rays, bit for bit, on two calls in a row — the second with a smaller input — plus the
size limit.  Collected last on purpose (the round's end runs pytest -x)."""
import ctypes as C

import numpy as np
import pytest

from rt_amd import World, _abi, scenes
from rt_amd._lib import RtError, check, lib

pytestmark = pytest.mark.gpu


def _rays(n, seed, lo=-8.0, hi=8.0):
    k = np.arange(n)
    o = np.stack([lo + (hi - lo) * scenes.hash_u01(seed, k, c) for c in range(3)], -1)
    o[:, 2] = 0.2 + 6.0 * scenes.hash_u01(seed, k, 7)
    d = np.stack([2.0 * scenes.hash_u01(seed + 1, k, c) - 1.0 for c in range(3)], -1)
    return np.ascontiguousarray(np.concatenate([o, d], -1)).view(_abi.ray_dt).reshape(-1)


def _noalloc(world, rays, t_min=0.001, t_max=1.7976931348623157e308):
    from hip_mem import DeviceBuffer, synchronize
    d_rays = DeviceBuffer.from_array(rays)
    d_hits = DeviceBuffer(len(rays) * _abi.hit_dt.itemsize)
    d_hits.fill(0xEE)                                   # stale bytes must all be overwritten
    check(lib().rt_trace_closest_noalloc(world._h, d_rays.ptr, len(rays), t_min, t_max, d_hits.ptr, None))
    synchronize()                                       # the entry point only queues the work
    out = d_hits.to_array(_abi.hit_dt, len(rays))
    d_rays.free()
    d_hits.free()
    return out


def test_noalloc_entry_matches_rt_trace_closest_twice_in_a_row():
    desc = scenes.mixed_test_scene()
    world = World.build(desc)
    big, small = _rays(200_003, 31), _rays(4_097, 32)
    for rays in (big, small):                           # the second call reuses the array the first one filled
        got = _noalloc(world, rays)
        ref = world.hit(rays)
        assert np.array_equal(got["prim"], ref["prim"]) and np.array_equal(got["t"], ref["t"])
        assert (got["_pad"] == 0).all()
        assert (got["prim"] != _abi.RT_MISS).mean() > 0.3
    got = _noalloc(world, small, 0.5, 7.0)
    ref = world.hit(small, 0.5, 7.0)
    assert np.array_equal(got["prim"], ref["prim"]) and np.array_equal(got["t"], ref["t"])
    # another scene (BVH4 with f32 boxes) on the same device shares the same array
    cover = World.build(scenes.cover_scene())
    got, ref = _noalloc(cover, small), cover.hit(small)
    assert np.array_equal(got["prim"], ref["prim"]) and np.array_equal(got["t"], ref["t"])


def test_noalloc_entry_leaves_device_memory_as_it_is():
    """The header promises no allocating call: free device memory (hipMemGetInfo) is the same before and after — also
    for the FIRST call on a scene, whose kernels rt_scene_create has run once so that no code object is loaded and no
    occupancy query is made here."""
    from hip_mem import DeviceBuffer, free_bytes, synchronize
    world = World.build(scenes.mixed_test_scene(seed=3, n_spheres=25, quads=9))      # a scene nothing has traced yet
    rays = _rays(70_001, 35)
    d_rays = DeviceBuffer.from_array(rays)
    d_hits = DeviceBuffer(len(rays) * _abi.hit_dt.itemsize)
    synchronize()
    before = free_bytes()
    for n in (len(rays), 1_000, len(rays)):
        check(lib().rt_trace_closest_noalloc(world._h, d_rays.ptr, n, 0.001, 1e300, d_hits.ptr, None))
        synchronize()
        assert free_bytes() == before
    got = d_hits.to_array(_abi.hit_dt, len(rays))
    ref = world.hit(rays, 0.001, 1e300)
    assert np.array_equal(got["prim"], ref["prim"]) and np.array_equal(got["t"], ref["t"])


def test_noalloc_entry_limits_and_arguments():
    from hip_mem import DeviceBuffer
    world = World.build(scenes.mixed_test_scene())
    L = lib()
    n_max = 1 << 18
    buf = DeviceBuffer(256)
    assert L.rt_trace_closest_noalloc(world._h, buf.ptr, n_max + 1, 0.001, 1e30, buf.ptr, None) == _abi.RT_ERR_LIMIT
    assert b"limit" in L.rt_last_error()
    assert L.rt_trace_closest_noalloc(world._h, None, 5, 0.001, 1e30, buf.ptr, None) == _abi.RT_ERR_INVALID
    assert L.rt_trace_closest_noalloc(None, buf.ptr, 5, 0.001, 1e30, buf.ptr, None) == _abi.RT_ERR_INVALID
    assert L.rt_trace_closest_noalloc(world._h, None, 0, 0.001, 1e30, None, None) == _abi.RT_OK
    rays = _rays(n_max, 33)                             # exactly the limit
    got = _noalloc(world, rays)
    ref = world.hit(rays)
    assert np.array_equal(got["prim"], ref["prim"]) and np.array_equal(got["t"], ref["t"])
