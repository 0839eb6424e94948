"""Loader for the CPU oracle (oracle/libhess_oracle.so) -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this module.
"""
import ctypes as C
import os
import subprocess

from hessgpu_amd import _abi
from hessgpu_amd.session import Session, make_params

_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_SO = os.path.join(_ROOT, "oracle", "libhess_oracle.so")
_lib = None
_fns = None


def build():
    """(Re)build the oracle with its Makefile (gcc; no GPU or reference needed)."""
    subprocess.run(["make", "-s", "-C", os.path.join(_ROOT, "oracle")], check=True)


def lib():
    global _lib, _fns
    if _lib is None:
        src = os.path.join(_ROOT, "oracle", "hess_oracle.c")
        if not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
            build()
        _lib = C.CDLL(_SO)
        _fns = _abi.bind(_lib, "hess_cpu_", _abi.PROTOTYPES)
        _lib.hess_cpu_create.restype = C.c_void_p  # (set/run_keypoints are bound through _abi.PROTOTYPES)
        _lib.hess_cpu_create.argtypes = [C.POINTER(_abi.HessParams)]
        _lib.hess_cpu_set_threads.argtypes = [C.c_void_p, C.c_int]
        _lib.hess_cpu_keep_levels.argtypes = [C.c_void_p, C.c_int]
        _lib.hess_cpu_filter_taps.restype = C.c_int
        _lib.hess_cpu_filter_taps.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_float)]
        _lib.hess_cpu_level_sigma.restype = C.c_float
        _lib.hess_cpu_level_sigma.argtypes = [C.c_void_p, C.c_int]
        _lib.hess_cpu_expf.restype = C.c_float
        _lib.hess_cpu_expf.argtypes = [C.c_float]
        _lib.hess_cpu_atan2f.restype = C.c_float
        _lib.hess_cpu_atan2f.argtypes = [C.c_float, C.c_float]
        _lib.hess_cpu_sincosf.argtypes = [C.c_float, C.POINTER(C.c_float), C.POINTER(C.c_float)]
        _lib.hess_cpu_f2h.restype = C.c_ushort
        _lib.hess_cpu_f2h.argtypes = [C.c_float]
        _lib.hess_cpu_h2f.restype = C.c_float
        _lib.hess_cpu_h2f.argtypes = [C.c_ushort]
    return _lib


class OracleSession(Session):
    def __init__(self, threads=1, keep_levels=True, **overrides):
        l = lib()
        detector = overrides.pop("detector", 0)   # the oracle's extension word (oracle/hess_oracle.h), not a product option
        border = overrides.pop("border", 0)       # analysis switch: 1 = the packed GLSL shaders' border rule
        p = make_params(_fns["default_params"], **overrides)
        p.reserved[0] = detector
        p.reserved[1] = border
        h = l.hess_cpu_create(C.byref(p))
        super().__init__(_fns, h, p)
        l.hess_cpu_set_threads(h, threads)
        l.hess_cpu_keep_levels(h, int(keep_levels))

    def filter_taps(self, level):
        taps = (C.c_float * 33)()
        n = lib().hess_cpu_filter_taps(self._h, level, taps)
        return [taps[i] for i in range(n)]

    def level_sigma(self, level):
        return lib().hess_cpu_level_sigma(self._h, level)


def oracle_quantize(desc):
    """float descriptors -> u8 as SiftMatchCU::SetDescriptors does (int(512*d + 0.5))."""
    import numpy as np

    L = lib()
    d = np.ascontiguousarray(desc, dtype=np.float32)
    out = np.zeros(d.shape, dtype=np.uint8)
    L.hess_cpu_match_quantize.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
    L.hess_cpu_match_quantize(d.ctypes.data, d.size, out.ctypes.data)
    return out


def oracle_match(des1, des2, loc1=None, loc2=None, H=None, F=None, distmax=0.7, ratiomax=0.8, hdistmax=32.0,
                 fdistmax=16.0, mutual_best=True, max_match=4096):
    """CPU oracle of the matcher (oracle/hess_oracle.c: hess_cpu_match); u8 descriptors [n,128]."""
    import numpy as np

    L = lib()
    L.hess_cpu_match.restype = C.c_int
    L.hess_cpu_match.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                 C.c_float, C.c_float, C.c_float, C.c_float, C.c_int, C.c_int, C.c_void_p]
    d1, d2 = np.ascontiguousarray(des1, np.uint8), np.ascontiguousarray(des2, np.uint8)
    arrs = [None if a is None else np.ascontiguousarray(a, np.float32) for a in (loc1, loc2, H, F)]
    ptr = [None if a is None else a.ctypes.data for a in arrs]
    out = np.zeros((max(max_match, 1), 2), dtype=np.int32)
    n = L.hess_cpu_match(d1.ctypes.data, len(d1), d2.ctypes.data, len(d2), ptr[0], ptr[1], ptr[2], ptr[3], distmax,
                         ratiomax, hdistmax, fdistmax, int(mutual_best), max_match, out.ctypes.data)
    return out[:n].copy()
