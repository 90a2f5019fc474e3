"""Device buffers for tests that hand DEVICE pointers to the C ABI, through the HIP runtime librt_hip.so already
loaded (ctypes on libamdhip64; no torch involved)."""
import ctypes as C

import numpy as np

_hip = None


def hip():
    global _hip
    if _hip is None:
        from rt_amd._lib import lib
        lib()                                         # loads librt_hip.so and with it the HIP runtime
        h = C.CDLL("libamdhip64.so.7", mode=C.RTLD_GLOBAL)
        h.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
        h.hipFree.argtypes = [C.c_void_p]
        h.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
        h.hipMemset.argtypes = [C.c_void_p, C.c_int, C.c_size_t]
        h.hipDeviceSynchronize.argtypes = []
        h.hipStreamCreate.argtypes = [C.POINTER(C.c_void_p)]
        h.hipStreamDestroy.argtypes = [C.c_void_p]
        h.hipStreamSynchronize.argtypes = [C.c_void_p]
        h.hipSetDevice.argtypes = [C.c_int]
        h.hipGetDevice.argtypes = [C.POINTER(C.c_int)]
        h.hipMemGetInfo.argtypes = [C.POINTER(C.c_size_t), C.POINTER(C.c_size_t)]
        _hip = h
    return _hip


def _ok(rc, what):
    if rc != 0:
        raise RuntimeError(f"{what} failed with HIP error {rc}")


class DeviceBuffer:
    def __init__(self, nbytes):
        self.nbytes = int(nbytes)
        self.ptr = C.c_void_p()
        _ok(hip().hipMalloc(C.byref(self.ptr), max(self.nbytes, 256)), "hipMalloc")

    @staticmethod
    def from_array(a):
        a = np.ascontiguousarray(a)
        b = DeviceBuffer(a.nbytes)
        if a.nbytes:
            _ok(hip().hipMemcpy(b.ptr, a.ctypes.data_as(C.c_void_p), a.nbytes, 1), "hipMemcpy H2D")
        return b

    def fill(self, byte):
        _ok(hip().hipMemset(self.ptr, byte, max(self.nbytes, 1)), "hipMemset")

    def to_array(self, dtype, count):
        out = np.zeros(count, dtype=dtype)
        assert out.nbytes <= self.nbytes
        if out.nbytes:
            _ok(hip().hipMemcpy(out.ctypes.data_as(C.c_void_p), self.ptr, out.nbytes, 2), "hipMemcpy D2H")
        return out

    def free(self):
        if self.ptr:
            hip().hipFree(self.ptr)
            self.ptr = C.c_void_p()

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


def synchronize():
    _ok(hip().hipDeviceSynchronize(), "hipDeviceSynchronize")


def free_bytes():
    """hipMemGetInfo's free device memory on the current device."""
    free, total = C.c_size_t(), C.c_size_t()
    _ok(hip().hipMemGetInfo(C.byref(free), C.byref(total)), "hipMemGetInfo")
    return free.value
