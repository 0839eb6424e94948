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
# the developer build of the same sources (-DHESS_DEV_SWITCHES: schedule A/B switches and test hooks read from the
# environment, csrc/hess_ctx.h); the tests that need a switch create their contexts from it
DEV_LIB_PATH = _os.path.join(_HERE, "dev", "libhessgpu.so")
_libs = {}   # dev? -> (CDLL, bound functions)


class HessLibraryMissing(ImportError):
    pass


def _load(dev):
    if dev not in _libs:
        path = DEV_LIB_PATH if dev else LIB_PATH
        if not _os.path.exists(path):
            raise HessLibraryMissing(
                f"{path} not found: build it with `python -m hessgpu_amd.build` "
                "(hipcc --offload-arch=gfx950). There is no CPU fallback.")
        # (RTLD_GLOBAL: the HIP runtime the library brings in must be the one PyTorch then finds, or a process ends up with
        # two runtimes and the second sees no GPU.  The two builds stay separate instances: they are linked -Bsymbolic)
        lib = _C.CDLL(path, mode=_C.RTLD_GLOBAL)
        table = dict(_abi.PROTOTYPES)
        table.update(_abi.PRODUCT_PROTOTYPES)
        fns = _abi.bind(lib, "hess_", table)
        lib.hess_math_probe.restype = _C.c_int
        lib.hess_math_probe.argtypes = [_C.c_void_p, _C.c_int, _C.c_void_p, _C.c_void_p, _C.c_void_p, _C.c_int]
        if dev and not fns["dev_switches"]():
            raise HessLibraryMissing(f"{path} is not a developer build (hess_dev_switches() == 0)")
        _libs[dev] = (lib, fns)
    return _libs[dev]


def load_library(dev=False):
    """dlopen libhessgpu.so (dev: the developer build) and bind every entry point of include/hess_abi.h (raises if absent)."""
    return _load(bool(dev))[0]


def functions(dev=False):
    """The bound entry points (name without the hess_ prefix -> ctypes function) of the product or the developer build."""
    return _load(bool(dev))[1]


def default_params(**overrides):
    return make_params(_load(False)[1]["default_params"], **overrides)


class HessContext(Session):
    """One hess_ctx on one HIP device (reference: one SiftGPU instance per device).  dev_switches=True: a context of
    the developer build, which reads the HESS_* schedule switches and test hooks from the environment."""

    def __init__(self, device=0, dev_switches=False, **overrides):
        lib, fns = _load(bool(dev_switches))
        p = make_params(fns["default_params"], **overrides)
        handle = fns["create"](device, _C.byref(p))
        if not handle:
            raise HessError(_abi.HESS_ERR_DEVICE,
                            f"hess_create failed on device {device} (no GPU visible or bad parameters)")
        super().__init__(fns, handle, p)
        self._lib = lib

    def math_probe(self, which, a, b=None):
        import numpy as np

        a = np.ascontiguousarray(a, dtype=np.float32)
        b = np.ascontiguousarray(b if b is not None else a, dtype=np.float32)
        out = np.zeros_like(a)
        self._check(self._lib.hess_math_probe(self._h, which, a.ctypes.data, b.ctypes.data, out.ctypes.data, a.size))
        return out
