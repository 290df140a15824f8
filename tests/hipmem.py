"""Device memory for tests through the HIP runtime the library itself is linked against (ctypes).  torch is not used for this: it
brings a runtime of its own, and a second runtime in a process that has used the card already finds no GPU."""
import ctypes as C

import numpy as np

from bamqc_amd import _lib


class Hip:
    """Device memory through the HIP runtime the library itself is linked against (ctypes; torch brings a runtime of its own, and a
    second one in a process that has used the card already finds no GPU)."""
    def __init__(self):
        _lib.load()
        self.rt = C.CDLL("libamdhip64.so", mode=C.RTLD_GLOBAL)
        self.rt.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
        self.rt.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
        self.rt.hipMemset.argtypes = [C.c_void_p, C.c_int, C.c_size_t]
        self.rt.hipFree.argtypes = [C.c_void_p]
        self.bufs = []

    def put(self, arr, extra=0, front=0):
        """the array in device memory, `front` zero bytes of the same allocation before it and `extra` (+ 256) behind it"""
        arr = np.ascontiguousarray(arr)
        p = C.c_void_p()
        assert self.rt.hipMalloc(C.byref(p), front + arr.nbytes + extra + 256) == 0
        assert self.rt.hipMemset(p, 0, front + arr.nbytes + extra + 256) == 0
        q = C.c_void_p(p.value + front)
        if arr.nbytes:
            assert self.rt.hipMemcpy(q, arr.ctypes.data, arr.nbytes, 1) == 0
        self.bufs.append(p)
        return q

    def get(self, p, nbytes, dtype=np.uint8):
        out = np.zeros(nbytes, np.uint8)
        if nbytes:
            assert self.rt.hipMemcpy(out.ctypes.data, p, nbytes, 2) == 0
        return out.view(dtype)

    def free(self):
        assert self.rt.hipDeviceSynchronize() == 0
        for p in self.bufs:
            self.rt.hipFree(p)
        self.bufs = []
