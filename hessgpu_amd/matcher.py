"""ctypes wrapper of the descriptor matcher entry points (hess_matcher_* in include/hess_abi.h):
the SiftMatchGPU surface of the reference (SetDescriptors / SetFeatureLocation / GetSiftMatch /
GetGuidedSiftMatch).  Plumbing only; the work happens in libhessgpu.so on the GPU."""
import ctypes as C

import numpy as np

from . import load_library
from .session import HessError

_bound = False


def _lib():
    global _bound
    L = load_library()
    if not _bound:
        L.hess_matcher_create.restype = C.c_void_p
        L.hess_matcher_create.argtypes = [C.c_int, C.c_int]
        L.hess_matcher_destroy.argtypes = [C.c_void_p]
        L.hess_matcher_set_max.argtypes = [C.c_void_p, C.c_int]
        L.hess_matcher_set_descriptors.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p]
        L.hess_matcher_set_descriptors_f32.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p]
        L.hess_matcher_set_locations.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int]
        L.hess_matcher_match.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_float,
                                         C.c_float, C.c_float, C.c_float, C.c_int]
        L.hess_matcher_last_ms.restype = C.c_float
        L.hess_matcher_last_ms.argtypes = [C.c_void_p]
        L.hess_matcher_last_error.restype = C.c_char_p
        L.hess_matcher_last_error.argtypes = [C.c_void_p]
        _bound = True
    return L


class Matcher:
    def __init__(self, device=0, max_sift=4096):
        self.L = _lib()
        self.h = self.L.hess_matcher_create(device, max_sift)
        if not self.h:
            raise HessError(-3, f"hess_matcher_create failed on device {device}")

    def close(self):
        if self.h:
            self.L.hess_matcher_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc):
        if rc < 0:
            raise HessError(rc, (self.L.hess_matcher_last_error(self.h) or b"").decode())
        return rc

    def set_descriptors(self, index, desc):
        d = np.ascontiguousarray(desc)
        if d.dtype == np.uint8:
            self._check(self.L.hess_matcher_set_descriptors(self.h, index, len(d), d.ctypes.data))
        else:
            d = np.ascontiguousarray(d, dtype=np.float32)
            self._check(self.L.hess_matcher_set_descriptors_f32(self.h, index, len(d), d.ctypes.data))

    def set_locations(self, index, loc, gap=0):
        a = np.ascontiguousarray(loc, dtype=np.float32)
        self._check(self.L.hess_matcher_set_locations(self.h, index, a.ctypes.data, gap))

    def match(self, max_match=4096, H=None, F=None, distmax=0.7, ratiomax=0.8, hdistmax=32.0, fdistmax=16.0,
              mutual_best=True):
        out = np.zeros((max(max_match, 1), 2), dtype=np.int32)
        h = np.ascontiguousarray(H, dtype=np.float32) if H is not None else None
        f = np.ascontiguousarray(F, dtype=np.float32) if F is not None else None
        n = self._check(self.L.hess_matcher_match(self.h, max_match, out.ctypes.data,
                                                  h.ctypes.data if h is not None else None,
                                                  f.ctypes.data if f is not None else None,
                                                  distmax, ratiomax, hdistmax, fdistmax, int(mutual_best)))
        return out[:n].copy()

    def last_ms(self):
        return float(self.L.hess_matcher_last_ms(self.h))
