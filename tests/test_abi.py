"""The C-ABI library loads, exports every symbol include/rt_hip.h declares, and fails loudly (never
falls back to a CPU path) when there is no GPU.  No compute calls here."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from rt_amd import _abi, _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_functions():
    src = open(os.path.join(ROOT, "include", "rt_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(rt_[a-z0-9_]+)\s*\(", src)))


def test_header_functions_are_exported_and_listed():
    names = declared_functions()
    assert set(names) == set(_lib.EXPORTS)
    L = _lib.lib()
    for n in names:
        assert hasattr(L, n), n


def test_abi_version_and_struct_sizes():
    assert _lib.lib().rt_abi_version() == 2
    hdr = open(os.path.join(ROOT, "rt_amd", "csrc", "rt_api.hip")).read()
    for name, (dt, size) in _abi.EXPECTED_SIZES.items():
        assert dt.itemsize == size
        assert f"sizeof({name}) == {size}" in hdr, name      # the C side static_asserts the same number


def test_invalid_arguments_are_rejected_before_touching_the_gpu():
    L = _lib.lib()
    sky = _abi.one(_abi.sky_dt)
    h = C.c_void_p()
    assert L.rt_scene_create(None, 1, None, 0, None, 0, None, 0, None, 0, _abi.ptr(sky), 0, C.byref(h)) == _abi.RT_ERR_INVALID
    assert b"NULL" in L.rt_last_error()
    s = _abi.one(_abi.sphere_dt)
    s["radius"], s["material"] = 1.0, 3
    assert L.rt_scene_create(_abi.ptr(s), 1, None, 0, None, 0, None, 0, None, 0, _abi.ptr(sky), 0, C.byref(h)) == _abi.RT_ERR_INVALID
    assert b"material" in L.rt_last_error()
    t = _abi.one(_abi.texture_dt)
    t["kind"] = _abi.TEX_CHECKER
    assert L.rt_scene_create(None, 0, None, 0, None, 0, _abi.ptr(t), 1, None, 0, _abi.ptr(sky), 0, C.byref(h)) == _abi.RT_ERR_INVALID
    t["kind"], t["width"], t["height"] = _abi.TEX_IMAGE, 4, 4
    assert L.rt_scene_create(None, 0, None, 0, None, 0, _abi.ptr(t), 1, None, 0, _abi.ptr(sky), 0, C.byref(h)) == _abi.RT_ERR_INVALID
    assert b"texel pool" in L.rt_last_error()
    L.rt_scene_destroy(None)      # NULL is a no-op
    assert L.rt_render(None, None, None, None, None) == _abi.RT_ERR_INVALID
    assert L.rt_trace_closest(None, None, 0, 0.0, 1.0, None, None) == _abi.RT_ERR_INVALID


def test_triangle_checks_report_the_first_offender_from_any_piece():
    """rt_scene_create validates triangles in pieces on the build threads: the triangle named in the message is the first
    bad one in the caller's order (what a plain loop reports), not whichever piece finished first."""
    L = _lib.lib()
    sky = _abi.one(_abi.sky_dt)
    h = C.c_void_p()
    n = 300_000                                     # several pieces of 65536
    tris = np.zeros(n, dtype=_abi.triangle_dt)
    tris["b"][:, 0] = 1.0
    tris["c"][:, 1] = 1.0
    mats = np.zeros(1, dtype=_abi.material_dt)
    mats["kind"] = 2                                # dielectric: needs no texture
    for bad, field in ((250_001, "material"), (70_000, "vertex")):
        if field == "material":
            tris["material"][bad] = 7
        else:
            tris["a"][bad, 2] = np.nan
        rc = L.rt_scene_create(None, 0, _abi.ptr(tris), n, _abi.ptr(mats), 1, None, 0, None, 0, _abi.ptr(sky), 0, C.byref(h))
        assert rc == _abi.RT_ERR_INVALID and not h.value
        msg = L.rt_last_error()
        assert (b"triangle %d:" % bad) in msg and (b"material" if field == "material" else b"non-finite") in msg, msg


def test_no_device_is_an_error_not_a_fallback():
    L = _lib.lib()
    n = C.c_int(-1)
    rc = L.rt_device_count(C.byref(n))
    if rc == 0 and n.value > 0:
        pytest.skip("a GPU is present")
    sky = _abi.one(_abi.sky_dt)
    h = C.c_void_p()
    rc = L.rt_scene_create(None, 0, None, 0, None, 0, None, 0, None, 0, _abi.ptr(sky), 0, C.byref(h))
    assert rc == _abi.RT_ERR_NO_DEVICE and not h.value
    assert b"no CPU fallback" in L.rt_last_error()
    from rt_amd import SceneDesc, World
    with pytest.raises(_lib.RtError):
        World.build(SceneDesc())


def test_gamma8_host_entry_point(oracle):
    from rt_amd import encode_gamma8
    c = np.linspace(-0.2, 1.3, 997)
    ref = np.zeros(len(c), np.uint8)
    oracle.lib().orc_gamma8(_abi.ptr(c), len(c), _abi.ptr(ref))
    assert np.array_equal(encode_gamma8(c), ref)
