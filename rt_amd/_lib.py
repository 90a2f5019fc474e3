"""ctypes loader for librt_hip.so (the product) — fails loudly, never falls back to a CPU path."""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# RT_HIP_LIB points at another build of the same library (A/B variants, an installed copy)
LIB_PATH = os.environ.get("RT_HIP_LIB") or os.path.join(_HERE, "librt_hip.so")

_lib = None


class RtError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"rt_hip error {code}: {msg}")
        self.code = code


def lib():
    """Load (once) and return the C-ABI library with argtypes set."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950). There is no CPU fallback.")
    L = C.CDLL(LIB_PATH, mode=C.RTLD_GLOBAL)
    vp, sz, u32, u64, i32, dbl = C.c_void_p, C.c_size_t, C.c_uint32, C.c_uint64, C.c_int, C.c_double
    L.rt_abi_version.restype = u32
    L.rt_abi_version.argtypes = []
    L.rt_last_error.restype = C.c_char_p
    L.rt_last_error.argtypes = []
    L.rt_device_count.argtypes = [C.POINTER(C.c_int)]
    L.rt_scene_create.argtypes = [vp, sz, vp, sz, vp, sz, vp, sz, vp, sz, vp, i32, C.POINTER(vp)]
    L.rt_scene_destroy.argtypes = [vp]
    L.rt_scene_destroy.restype = None
    L.rt_scene_get_info.argtypes = [vp, vp]
    L.rt_trace_closest.argtypes = [vp, vp, sz, dbl, dbl, vp, vp]
    L.rt_trace_closest_noalloc.argtypes = [vp, vp, sz, dbl, dbl, vp, vp]
    L.rt_trace_primary.argtypes = [vp, vp, u32, i32, u64, vp, vp]
    L.rt_render.argtypes = [vp, vp, vp, vp, vp]
    L.rt_render_accumulate_device.argtypes = [vp, vp, vp, vp, vp, vp]
    L.rt_scene_sync.argtypes = [vp, vp]
    L.rt_resolve_device.argtypes = [vp, vp, sz, u64, i32, vp]
    L.rt_encode_gamma8.argtypes = [vp, sz, vp]
    L.rt_encode_gamma8_device.argtypes = [vp, sz, vp, i32, vp]
    L.rt_render_pass.argtypes = [vp, vp, u32, u64, u64, u32, vp, vp, vp]
    L.rt_render_multi.argtypes = [vp, u32, vp, vp, vp, vp]
    for name in EXPORTS:
        if name not in ("rt_abi_version", "rt_last_error", "rt_scene_destroy"):
            getattr(L, name).restype = C.c_int
    _lib = L
    return L


# every symbol include/rt_hip.h declares
EXPORTS = [
    "rt_abi_version", "rt_last_error", "rt_device_count", "rt_scene_create", "rt_scene_destroy",
    "rt_scene_get_info", "rt_trace_closest", "rt_trace_closest_noalloc", "rt_trace_primary", "rt_render",
    "rt_render_accumulate_device", "rt_scene_sync", "rt_resolve_device", "rt_encode_gamma8", "rt_encode_gamma8_device",
    "rt_render_pass", "rt_render_multi",
]


def check(code):
    if code != 0:
        raise RtError(code, lib().rt_last_error().decode("utf-8", "replace"))
