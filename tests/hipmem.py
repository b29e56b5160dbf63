"""Minimal device-memory helper for the GPU tests (ctypes on the HIP runtime the product
library already loaded) -- keeps the parity tests independent of torch."""
import ctypes as C

import numpy as np

import urban_road_filter_amd as u

_hip = None


def hip():
    global _hip
    if _hip is None:
        u.lib()   # make sure ONE runtime is in the process, then bind to it by soname
        _hip = C.CDLL("libamdhip64.so.7")
        _hip.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
        _hip.hipFree.argtypes = [C.c_void_p]
        _hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
        _hip.hipMemset.argtypes = [C.c_void_p, C.c_int, C.c_size_t]
        _hip.hipDeviceSynchronize.argtypes = []
    return _hip


class DevBuf:
    def __init__(self, nbytes):
        self.nbytes = int(nbytes)
        self.p = C.c_void_p()
        rc = hip().hipMalloc(C.byref(self.p), max(self.nbytes, 1))
        assert rc == 0, "hipMalloc failed: %d" % rc
        self.ptr = self.p.value

    @classmethod
    def from_numpy(cls, a):
        a = np.ascontiguousarray(a)
        b = cls(a.nbytes)
        if a.nbytes:
            assert hip().hipMemcpy(b.ptr, a.ctypes.data, a.nbytes, 1) == 0
        return b

    def fill(self, byte):
        assert hip().hipMemset(self.ptr, byte, self.nbytes) == 0
        # hipMemset runs on the null stream; the library's streams are non-blocking, i.e. NOT ordered behind it: with sixteen fuzz
        # processes on one GPU the fill has been seen to land after the kernels that wrote the buffer
        assert hip().hipDeviceSynchronize() == 0

    def to_numpy(self, dtype, count=None):
        dtype = np.dtype(dtype)
        count = self.nbytes // dtype.itemsize if count is None else count
        out = np.empty(count, dtype)
        assert hip().hipDeviceSynchronize() == 0
        if out.nbytes:
            assert hip().hipMemcpy(out.ctypes.data, self.ptr, out.nbytes, 2) == 0
        return out

    def data_ptr(self):
        return self.ptr

    def free(self):
        if self.p:
            hip().hipFree(self.p)
            self.p = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass
