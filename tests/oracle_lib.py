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
        p = make_params(_fns["default_params"], **overrides)
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
