"""hessgpu_amd -- MI355X-native Hessian interest points + SIFT descriptors (HessGPU hot path).

The product is the native library `libhessgpu.so` (hand-written HIP kernels for gfx950 behind the
C ABI of include/hess_abi.h) and `libsiftgpu.so` (the SiftGPU C++ plugin surface on top of it).
This package is host-side plumbing: it loads the library, mirrors the ABI with ctypes and shards
image batches over ranks with torch.distributed.  There is no CPU fallback: if the HIP library is
missing or no GPU is usable, creating a context raises.
"""
import ctypes as _C
import os as _os

from . import _abi
from .session import HessError, Session, make_params

_HERE = _os.path.dirname(_os.path.abspath(__file__))
LIB_PATH = _os.environ.get("HESS_LIB") or _os.path.join(_HERE, "libhessgpu.so")  # HESS_LIB: developer override
_lib = None
_fns = None


class HessLibraryMissing(ImportError):
    pass


def load_library():
    """dlopen libhessgpu.so and bind every entry point of include/hess_abi.h (raises if absent)."""
    global _lib, _fns
    if _lib is None:
        if not _os.path.exists(LIB_PATH):
            raise HessLibraryMissing(
                f"{LIB_PATH} not found: build it with `python -m hessgpu_amd.build` "
                "(hipcc --offload-arch=gfx950). There is no CPU fallback.")
        lib = _C.CDLL(LIB_PATH, mode=_C.RTLD_GLOBAL)
        table = dict(_abi.PROTOTYPES)
        table.update(_abi.PRODUCT_PROTOTYPES)
        fns = _abi.bind(lib, "hess_", table)
        lib.hess_math_probe.restype = _C.c_int
        lib.hess_math_probe.argtypes = [_C.c_void_p, _C.c_int, _C.c_void_p, _C.c_void_p, _C.c_void_p, _C.c_int]
        _lib, _fns = lib, fns
    return _lib


def default_params(**overrides):
    load_library()
    return make_params(_fns["default_params"], **overrides)


class HessContext(Session):
    """One hess_ctx on one HIP device (reference: one SiftGPU instance per device)."""

    def __init__(self, device=0, **overrides):
        lib = load_library()
        p = make_params(_fns["default_params"], **overrides)
        handle = _fns["create"](device, _C.byref(p))
        if not handle:
            raise HessError(_abi.HESS_ERR_DEVICE,
                            f"hess_create failed on device {device} (no GPU visible or bad parameters)")
        super().__init__(_fns, handle, p)
        self._lib = lib

    def math_probe(self, which, a, b=None):
        import numpy as np

        a = np.ascontiguousarray(a, dtype=np.float32)
        b = np.ascontiguousarray(b if b is not None else a, dtype=np.float32)
        out = np.zeros_like(a)
        self._check(self._lib.hess_math_probe(self._h, which, a.ctypes.data, b.ctypes.data, out.ctypes.data, a.size))
        return out
