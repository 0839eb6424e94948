"""ctypes mirror of include/hess_abi.h (the C-ABI drop-in boundary).

Plumbing only: structure layouts, enum values and the prototype table.  `bind(lib, prefix)`
attaches the prototypes to a loaded shared library whose entry points are named
`<prefix><name>` -- `hess_` for the product library (libhessgpu.so).  The parity tests bind the
CPU oracle's `hess_cpu_` entry points with the same table so both sides read alike.
"""
import ctypes as C

HESS_ABI_VERSION = 4

HESS_OK = 0
HESS_ERR_ARG = -1
HESS_ERR_TOO_BIG = -2
HESS_ERR_DEVICE = -3
HESS_ERR_NOMEM = -4
HESS_ERR_STATE = -5
HESS_ERR_UNSUPPORTED = -6

TYPE_DARK_BLOB, TYPE_BRIGHT_BLOB, TYPE_SADDLE, TYPE_NONE = 0, 1, 2, 3
TRUNC_HIGHEST_0, TRUNC_HIGHEST_1, TRUNC_LOWEST, TRUNC_TOPK = 0, 1, 2, 3
DESC_ORDER_INTERLEAVED, DESC_ORDER_SEQUENTIAL, DESC_ORDER_PIXEL = 0, 1, 2
FMT_LUM, FMT_LUM_ALPHA, FMT_RGB, FMT_RGBA, FMT_BGR, FMT_BGRA = 1, 2, 3, 4, 5, 6
PIX_U8, PIX_U16, PIX_F32 = 1, 2, 3
DBG_GAUSS, DBG_DETH, DBG_GOT = 0, 1, 2
(T_LOAD, T_ALLOC, T_PYRAMID, T_DETECT, T_LIST, T_ORIENT, T_MULTI_ORIENT, T_DOWNLOAD,
 T_DESCRIPTOR, T_VBO, T_REDUCTION, T_TOTAL, T_COUNT) = range(13)
(K_GAUSS, K_DOWNSAMPLE, K_HESSIAN, K_EXTREMA, K_TOPK, K_ORIENT, K_DESCRIPTOR, K_INPUT,
 K_GAUSS_OCT0, K_COUNT) = range(10)
KERNEL_NAMES = ["gauss", "downsample", "hessian", "extrema", "topk", "orient", "descriptor",
                "input", "gauss_octave0"]  # gauss_octave0 repeats the octave-0 launches counted in gauss


class HessParams(C.Structure):
    _fields_ = [
        ("abi_version", C.c_int32),
        ("dog_level_num", C.c_int32),
        ("sigma0", C.c_float),
        ("sigman", C.c_float),
        ("dog_threshold", C.c_float),
        ("edge_threshold", C.c_float),
        ("filter_width_factor", C.c_float),
        ("orient_window_factor", C.c_float),
        ("orient_gaussian_factor", C.c_float),
        ("desc_window_factor", C.c_float),
        ("first_octave", C.c_int32),
        ("octave_num", C.c_int32),
        ("subpixel", C.c_int32),
        ("max_orientation", C.c_int32),
        ("fixed_orientation", C.c_int32),
        ("lowe_origin", C.c_int32),
        ("half_sift", C.c_int32),
        ("compute_descriptors", C.c_int32),
        ("normalize", C.c_int32),
        ("truncate_method", C.c_int32),
        ("feature_count_threshold", C.c_int32),
        ("tex_max_dim", C.c_int32),
        ("auto_downscale", C.c_int32),
        ("verbose", C.c_int32),
        ("dynamic_indexing", C.c_int32),
        ("descriptor_order", C.c_int32),   # DESC_ORDER_*
        ("reserved", C.c_int32 * 6),   # must be zero for the product (word 0: the test oracle's detector switch)
    ]


class HessKeypoint(C.Structure):
    _fields_ = [("x", C.c_float), ("y", C.c_float), ("s", C.c_float), ("o", C.c_float),
                ("response", C.c_float), ("level", C.c_uint16), ("type", C.c_uint16)]


class HessRawKey(C.Structure):
    _fields_ = [("level_index", C.c_int32), ("col", C.c_int32), ("row", C.c_int32),
                ("packed", C.c_uint32), ("dx", C.c_float), ("dy", C.c_float), ("ds", C.c_float),
                ("pad", C.c_uint32)]


assert C.sizeof(HessKeypoint) == 24
assert C.sizeof(HessRawKey) == 32

# numpy dtypes with the same layout
import numpy as _np

KEYPOINT_DTYPE = _np.dtype([("x", "<f4"), ("y", "<f4"), ("s", "<f4"), ("o", "<f4"),
                            ("response", "<f4"), ("level", "<u2"), ("type", "<u2")])
RAWKEY_DTYPE = _np.dtype([("level_index", "<i4"), ("col", "<i4"), ("row", "<i4"),
                          ("packed", "<u4"), ("dx", "<f4"), ("dy", "<f4"), ("ds", "<f4"),
                          ("pad", "<u4")])
assert KEYPOINT_DTYPE.itemsize == 24 and RAWKEY_DTYPE.itemsize == 32

_ctx = C.c_void_p
_P = C.POINTER

# name -> (restype, argtypes, takes_device)
# Entry points every implementation of the ABI has (product: all; oracle: those it implements).
PROTOTYPES = {
    "default_params": (None, [_P(HessParams)]),
    "destroy": (None, [_ctx]),
    "run_host": (C.c_int, [_ctx, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_size_t, C.c_int,
                           C.c_int, C.c_int]),
    "set_keypoints": (C.c_int, [_ctx, C.c_void_p, C.c_int, C.c_int]),
    "run_keypoints": (C.c_int, [_ctx, C.c_void_p, C.c_int, C.c_int]),
    "count": (C.c_int, [_ctx, C.c_int]),
    "desc_dim": (C.c_int, [_ctx]),
    "fetch": (C.c_int, [_ctx, C.c_int, C.c_void_p, C.c_void_p]),
    "geometry": (C.c_int, [_ctx, _P(C.c_int), _P(C.c_int)]),
    "debug_level": (C.c_int, [_ctx, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "debug_list": (C.c_int, [_ctx, C.c_int, C.c_void_p, C.c_int]),
    "debug_key_levels": (C.c_int, [_ctx, C.c_void_p, C.c_int]),
    "timing": (_P(C.c_float), [_ctx]),
    "last_error": (C.c_char_p, [_ctx]),
}
# Product-only entry points.
PRODUCT_PROTOTYPES = {
    "create": (_ctx, [C.c_int, _P(HessParams)]),
    "reserve": (C.c_int, [_ctx, C.c_int, C.c_int, C.c_int]),
    "run_device": (C.c_int, [_ctx, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_size_t, C.c_int,
                             C.c_int, C.c_int]),
    "submit_device": (C.c_int, [_ctx, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_size_t, C.c_int,
                                C.c_int, C.c_int]),
    "submit_host": (C.c_int, [_ctx, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_size_t, C.c_int,
                              C.c_int, C.c_int]),
    "wait": (C.c_int, [_ctx]),
    "device_results": (C.c_int, [_ctx, _P(C.c_void_p), _P(C.c_void_p), _P(C.c_int)]),
    "profile_enable": (C.c_int, [_ctx, C.c_int]),
    "profile_get": (C.c_int, [_ctx, C.c_int, _P(C.c_double), _P(C.c_longlong), _P(C.c_double)]),
    "profile_reset": (C.c_int, [_ctx]),
    "profile_get_in_lds": (C.c_int, [_ctx, C.c_int, C.POINTER(C.c_double)]),
    "debug_regrown": (C.c_int, [_ctx]),
    "debug_keep_levels": (C.c_int, [_ctx, C.c_int]),
    "last_input": (C.c_int, [_ctx, C.c_void_p, C.c_size_t]),
    "share_results": (C.c_int, [_ctx, C.c_char_p]),
    "shared_results_info": (C.c_int, [_ctx, _P(C.c_uint), _P(C.c_uint), _P(C.c_size_t), _P(C.c_size_t)]),
    "dev_switches": (C.c_int, []),
}


def bind(lib, prefix, table):
    """Attach prototypes; raises AttributeError naming the first missing symbol."""
    out = {}
    for name, (res, args) in table.items():
        fn = getattr(lib, prefix + name)  # AttributeError if the symbol is not exported
        fn.restype = res
        fn.argtypes = args
        out[name] = fn
    return out
