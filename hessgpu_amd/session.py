"""Thin object wrapper over one implementation of the C ABI (include/hess_abi.h).

`Session` owns one `hess_ctx*` and exposes the calls the parity tests and bench need with numpy
arrays at the edge.  It contains no arithmetic: every result comes out of the bound library.
"""
import ctypes as C

import numpy as np

from . import _abi


class HessError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"hess error {code}: {msg}")
        self.code = code


_FMT_BY_CHANNELS = {1: _abi.FMT_LUM, 2: _abi.FMT_LUM_ALPHA, 3: _abi.FMT_RGB, 4: _abi.FMT_RGBA}
_PIX_BY_DTYPE = {np.dtype(np.uint8): _abi.PIX_U8, np.dtype(np.uint16): _abi.PIX_U16,
                 np.dtype(np.float32): _abi.PIX_F32}


def make_params(fn_default, **overrides):
    p = _abi.HessParams()
    fn_default(C.byref(p))
    for k, v in overrides.items():
        if not hasattr(p, k):
            raise AttributeError(f"hess_params has no field {k!r}")
        setattr(p, k, v)
    return p


class Session:
    """One context of one backend.  `fns` is the dict produced by _abi.bind()."""

    def __init__(self, fns, handle, params):
        self._f = fns
        self._h = handle
        self.params = params
        self._batch = 0
        if not handle:
            raise HessError(_abi.HESS_ERR_DEVICE, "context creation failed")

    def close(self):
        if self._h:
            self._f["destroy"](self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- helpers -------------------------------------------------------------------------
    def _check(self, rc):
        if rc < 0:
            msg = self._f["last_error"](self._h)
            raise HessError(rc, msg.decode() if msg else "")
        return rc

    @staticmethod
    def _describe(images, fmt):
        a = np.ascontiguousarray(images)
        if a.dtype not in _PIX_BY_DTYPE:
            raise TypeError(f"unsupported pixel dtype {a.dtype}")
        if a.ndim == 2:
            a = a[None]
        if a.ndim == 3:  # [B,H,W] luminance
            nch = 1
        elif a.ndim == 4:
            nch = a.shape[3]
        else:
            raise ValueError("images must be [H,W], [B,H,W] or [B,H,W,C]")
        b, h, w = a.shape[:3]
        fmt = fmt or _FMT_BY_CHANNELS[nch]
        pitch = w * nch * a.dtype.itemsize
        return a, b, h, w, pitch, pitch * h, fmt, _PIX_BY_DTYPE[a.dtype]

    # -- the path ------------------------------------------------------------------------
    def run(self, images, fmt=None):
        """Host pixels -> features (SiftGPU::RunSIFT(w,h,data,fmt,type) for a batch)."""
        a, b, h, w, pitch, stride, fmt, pix = self._describe(images, fmt)
        self._check(self._f["run_host"](self._h, a.ctypes.data_as(C.c_void_p), w, h, pitch, stride,
                                        b, fmt, pix))
        self._batch = b
        return [self.count(i) for i in range(b)]

    def run_device(self, dev_ptr, batch, height, width, channels=1, pixtype=_abi.PIX_U8, fmt=None):
        """Pixels already in HBM (raw device pointer, e.g. torch tensor .data_ptr())."""
        fmt = fmt or _FMT_BY_CHANNELS[channels]
        isz = {_abi.PIX_U8: 1, _abi.PIX_U16: 2, _abi.PIX_F32: 4}[pixtype]
        pitch = width * channels * isz
        self._check(self._f["run_device"](self._h, C.c_void_p(dev_ptr), width, height, pitch,
                                          pitch * height, batch, fmt, pixtype))
        self._batch = batch

    def submit_device(self, dev_ptr, batch, height, width, channels=1, pixtype=_abi.PIX_U8, fmt=None):
        """Asynchronous half of run_device: enqueue on the context's stream and return."""
        fmt = fmt or _FMT_BY_CHANNELS[channels]
        isz = {_abi.PIX_U8: 1, _abi.PIX_U16: 2, _abi.PIX_F32: 4}[pixtype]
        pitch = width * channels * isz
        self._check(self._f["submit_device"](self._h, C.c_void_p(dev_ptr), width, height, pitch,
                                             pitch * height, batch, fmt, pixtype))
        self._batch = batch

    def submit_host(self, images=None, fmt=None, ptr=None, batch=None, height=None, width=None):
        """Asynchronous half of run(): host pixels (a numpy array, or a raw host pointer `ptr` to `batch` u8
        luminance images, e.g. a pinned torch tensor's .data_ptr()) -> enqueue transfer + path, return."""
        if ptr is not None:
            pitch = width
            self._check(self._f["submit_host"](self._h, C.c_void_p(ptr), width, height, pitch, pitch * height, batch,
                                               _abi.FMT_LUM, _abi.PIX_U8))
            self._batch = batch
            return
        a, b, h, w, pitch, stride, fmt, pix = self._describe(images, fmt)
        self._check(self._f["submit_host"](self._h, a.ctypes.data_as(C.c_void_p), w, h, pitch, stride, b, fmt, pix))
        self._batch = b

    def wait(self):
        """Block until the submitted batch's keypoints and descriptors are in host memory."""
        self._check(self._f["wait"](self._h))

    def set_keypoints(self, keys, have_orientation=True):
        """SiftGPU::SetKeypointList: the next run() of ONE image uses these keypoints, no detection."""
        k = np.ascontiguousarray(keys, dtype=_abi.KEYPOINT_DTYPE)
        self._check(self._f["set_keypoints"](self._h, k.ctypes.data_as(C.c_void_p), len(k), int(have_orientation)))

    def run_keypoints(self, keys, have_orientation=True):
        """SiftGPU::RunSIFT(num, keys, flag): orientation/descriptors for `keys` on the current image."""
        k = np.ascontiguousarray(keys, dtype=_abi.KEYPOINT_DTYPE)
        self._check(self._f["run_keypoints"](self._h, k.ctypes.data_as(C.c_void_p), len(k), int(have_orientation)))
        return self.count(0)

    def debug_key_levels(self, levels=None):
        """Parity hook: explicit level index per user keypoint for the following set/run_keypoints (None clears)."""
        if levels is None:
            self._check(self._f["debug_key_levels"](self._h, None, 0))
            return
        lv = np.ascontiguousarray(levels, dtype=np.int32)
        self._check(self._f["debug_key_levels"](self._h, lv.ctypes.data_as(C.c_void_p), len(lv)))

    def reserve(self, width, height, batch):
        self._check(self._f["reserve"](self._h, width, height, batch))

    def count(self, img=0):
        return self._check(self._f["count"](self._h, img))

    def desc_dim(self):
        return self._check(self._f["desc_dim"](self._h))

    def fetch(self, img=0):
        """-> (keys structured array [N], descriptors float32 [N, dim])."""
        n = self.count(img)
        dim = self.desc_dim()
        keys = np.zeros(n, dtype=_abi.KEYPOINT_DTYPE)
        desc = np.zeros((n, dim), dtype=np.float32)
        self._check(self._f["fetch"](self._h, img, keys.ctypes.data_as(C.c_void_p),
                                     desc.ctypes.data_as(C.c_void_p) if dim else None))
        return keys, desc

    def geometry(self):
        ws = (C.c_int * 32)()
        hs = (C.c_int * 32)()
        n = self._check(self._f["geometry"](self._h, ws, hs))
        return [(ws[i], hs[i]) for i in range(n)]

    def level(self, img, octave, level, what):
        w, h = self.geometry()[octave]
        n = w * h * (2 if what == _abi.DBG_GOT else 1)
        out = np.zeros(n, dtype=np.float32)
        self._check(self._f["debug_level"](self._h, img, octave, level, what,
                                           out.ctypes.data_as(C.c_void_p)))
        return out.reshape(h, w, 2) if what == _abi.DBG_GOT else out.reshape(h, w)

    def rawlist(self, img=0):
        n = self._check(self._f["debug_list"](self._h, img, None, 0))
        out = np.zeros(n, dtype=_abi.RAWKEY_DTYPE)
        if n:
            self._check(self._f["debug_list"](self._h, img, out.ctypes.data_as(C.c_void_p), n))
        return out

    def timing(self):
        p = self._f["timing"](self._h)
        return np.array([p[i] for i in range(_abi.T_COUNT)], dtype=np.float32)

    def device_results(self):
        """-> (keys_ptr, desc_ptr, total): packed device results of the last run (product only)."""
        k, d, cap = C.c_void_p(), C.c_void_p(), C.c_int()
        self._check(self._f["device_results"](self._h, C.byref(k), C.byref(d), C.byref(cap)))
        return k.value, d.value, cap.value

    def keep_levels(self, on=True):
        """Product: also store the top Gaussian level of every octave (never materialised by default) so that
        level(..., DBG_GAUSS) can return it; the test oracle keeps all levels anyway."""
        if "debug_keep_levels" in self._f:
            self._check(self._f["debug_keep_levels"](self._h, int(on)))

    def regrown(self):
        """Times the context grew its feature storage after an overflow and ran the batch again (product only)."""
        return self._check(self._f["debug_regrown"](self._h))

    # -- node-shared result buffers (product only; hess_abi.h, hess_share_results) --------
    def share_results(self, name):
        """Keep this context's pinned result buffers in POSIX shared memory objects "/<name>.h|.k<n>|.d<n>" so that
        another process of the node reads them in place (dist.SharedResultsReader).  Before the first batch."""
        self._check(self._f["share_results"](self._h, name.encode()))

    def shared_results_info(self):
        """-> (gen_keys, gen_desc, keys_bytes, desc_bytes) of the shared result buffers."""
        gk, gd, kb, db = C.c_uint(), C.c_uint(), C.c_size_t(), C.c_size_t()
        self._check(self._f["shared_results_info"](self._h, C.byref(gk), C.byref(gd), C.byref(kb), C.byref(db)))
        return gk.value, gd.value, kb.value, db.value

    # -- profiling (product only) --------------------------------------------------------
    def profile_enable(self, on=True):
        self._check(self._f["profile_enable"](self._h, int(on)))

    def profile_reset(self):
        self._check(self._f["profile_reset"](self._h))

    def profile(self):
        out = {}
        for k, name in enumerate(_abi.KERNEL_NAMES):
            ms, n, by = C.c_double(), C.c_longlong(), C.c_double()
            self._check(self._f["profile_get"](self._h, k, C.byref(ms), C.byref(n), C.byref(by)))
            out[name] = {"ms": ms.value, "launches": n.value, "bytes": by.value}
            if "profile_get_in_lds" in self._f:   # bytes of the reference's layout that never left LDS (hess_abi.h)
                il = C.c_double()
                self._check(self._f["profile_get_in_lds"](self._h, k, C.byref(il)))
                out[name]["bytes_in_lds"] = il.value
        return out
