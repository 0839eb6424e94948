"""ctypes access to libsiftgpu.so (the SiftGPU C++ class) through its flat C mirror."""
import ctypes as C
import os

import numpy as np

from hessgpu_amd import _abi

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(os.path.dirname(_HERE), "hessgpu_amd", "libsiftgpu.so")
GL_LUMINANCE, GL_RGB, GL_RGBA, GL_BGR = 0x1909, 0x1907, 0x1908, 0x80E0
GL_UNSIGNED_BYTE, GL_UNSIGNED_SHORT, GL_FLOAT = 0x1401, 0x1403, 0x1406
_lib = None


def lib():
    global _lib
    if _lib is None:
        L = C.CDLL(_SO)
        L.CreateNewSiftGPU.restype = C.c_void_p
        L.CreateNewSiftGPU.argtypes = [C.c_int]
        L.CreateNewSiftMatchGPU.restype = C.c_void_p
        L.CreateComboSiftGPU.restype = C.c_void_p
        L.CreateRemoteSiftGPU.restype = C.c_void_p
        for name, res, args in [
            ("siftgpu_destroy", None, [C.c_void_p]),
            ("siftgpu_parse_param", None, [C.c_void_p, C.c_int, C.POINTER(C.c_char_p)]),
            ("siftgpu_create_context", C.c_int, [C.c_void_p]),
            ("siftgpu_run_data", C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_uint, C.c_uint]),
            ("siftgpu_run_file", C.c_int, [C.c_void_p, C.c_char_p]),
            ("siftgpu_run_index", C.c_int, [C.c_void_p, C.c_int]),
            ("siftgpu_feature_num", C.c_int, [C.c_void_p]),
            ("siftgpu_feature_vector", None, [C.c_void_p, C.c_void_p, C.c_void_p]),
            ("siftgpu_save", None, [C.c_void_p, C.c_char_p]),
            ("siftgpu_timing", C.POINTER(C.c_float), [C.c_void_p]),
            ("siftgpu_set_verbose", None, [C.c_void_p, C.c_int]),
            ("siftgpu_image_count", C.c_int, [C.c_void_p]),
            ("siftgpu_get_params", C.c_int, [C.c_void_p, C.c_void_p]),
            ("siftgpu_descriptor_dim", C.c_int, [C.c_void_p]),
        ]:
            fn = getattr(L, name)
            fn.restype, fn.argtypes = res, args
        _lib = L
    return _lib


class SiftGPU:
    def __init__(self, args=()):
        self.L = lib()
        self.h = self.L.CreateNewSiftGPU(1)
        self.parse(["-v", "0"] + list(args))

    def parse(self, args):
        arr = (C.c_char_p * len(args))(*[a.encode() for a in args])
        self.L.siftgpu_parse_param(self.h, len(args), arr)

    def params(self):
        p = _abi.HessParams()
        assert self.L.siftgpu_get_params(self.h, C.byref(p)) == 0
        return p

    def create_context(self):
        return self.L.siftgpu_create_context(self.h)

    def run(self, img, gl_format, gl_type):
        a = np.ascontiguousarray(img)
        return self.L.siftgpu_run_data(self.h, a.shape[1], a.shape[0], a.ctypes.data, gl_format, gl_type)

    def run_file(self, path):
        return self.L.siftgpu_run_file(self.h, path.encode())

    def features(self):
        n = self.L.siftgpu_feature_num(self.h)
        dim = self.L.siftgpu_descriptor_dim(self.h)
        keys = np.zeros(n, dtype=_abi.KEYPOINT_DTYPE)
        desc = np.zeros((n, max(dim, 0)), dtype=np.float32)
        self.L.siftgpu_feature_vector(self.h, keys.ctypes.data, desc.ctypes.data if dim > 0 else None)
        return keys, desc

    def save(self, path):
        self.L.siftgpu_save(self.h, path.encode())

    def timing(self):
        t = self.L.siftgpu_timing(self.h)
        return [t[i] for i in range(12)]

    def close(self):
        if self.h:
            self.L.siftgpu_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
