"""The SiftGPU C++ plugin surface (libsiftgpu.so) without a device: argv parsing semantics of
SiftGPU::ParseParam (reference SiftGPU.cpp:855-1380) and the exported factories."""
import ctypes as C

import pytest

import siftgpu_lib
from hessgpu_amd import _abi


def _params(args):
    s = siftgpu_lib.SiftGPU(args)
    p = s.params()
    s.close()
    return p


def test_factories_exist_and_unsupported_ones_return_null():
    L = siftgpu_lib.lib()
    for name in ("CreateNewSiftGPU", "CreateNewSiftMatchGPU", "CreateComboSiftGPU", "CreateRemoteSiftGPU"):
        assert hasattr(L, name)
    L.CreateRemoteSiftGPU.argtypes = [C.c_int, C.c_char_p]
    assert L.CreateRemoteSiftGPU(7777, None) is None  # TCP server mode is out of scope
    assert L.CreateComboSiftGPU() is not None and L.CreateNewSiftMatchGPU(4096) is not None


def test_defaults_resolve_like_ParseSiftParam():
    p = _params([])
    assert p.dog_level_num == 3 and abs(p.sigma0 - 1.6) < 1e-7 and abs(p.dog_threshold - 0.02 / 3) < 1e-9
    assert p.edge_threshold == 10.0 and p.max_orientation == 2 and p.subpixel == 1
    assert p.truncate_method == _abi.TRUNC_HIGHEST_0 and p.feature_count_threshold == -1


def test_value_options():
    p = _params(["-t", "0.01", "-e", "5", "-d", "4", "-fo", "1", "-no", "3", "-f", "5.0", "-w", "1.5", "-dw", "2.5",
                 "-maxd", "4096"])
    assert abs(p.dog_threshold - 0.01) < 1e-9 and p.edge_threshold == 5.0 and p.dog_level_num == 4
    assert p.first_octave == 1 and p.octave_num == 3 and p.tex_max_dim == 4096
    assert p.filter_width_factor == 5.0 and p.orient_window_factor == 1.5 and p.desc_window_factor == 2.5


def test_out_of_range_values_are_ignored():
    assert _params([]).dynamic_indexing == 0 and _params(["-di"]).dynamic_indexing == 1  # SiftGPU.cpp:1030
    p = _params(["-t", "0.7", "-e", "-1", "-d", "11", "-fo", "-1", "-topk", "0"])
    assert abs(p.dog_threshold - 0.02 / 3) < 1e-9 and p.edge_threshold == 10.0 and p.dog_level_num == 3
    assert p.first_octave == 0
    assert p.truncate_method == _abi.TRUNC_TOPK and p.feature_count_threshold == -1  # method set, count not


def test_flags_and_truncation_methods():
    p = _params(["-half", "-sd", "-ads", "-loweo", "-ofix"])
    assert p.half_sift == 1 and p.compute_descriptors == 0 and p.auto_downscale == 1 and p.lowe_origin == 1
    assert p.fixed_orientation == 1
    assert _params(["-ofix", "-ofix-not"]).fixed_orientation == 0
    for opt, method in (("-tc", 0), ("-tc1", 0), ("-tc2", 1), ("-tc3", 2), ("-topk", 3)):
        p = _params([opt, "500"])
        assert p.truncate_method == method and p.feature_count_threshold == 500


def test_first_four_characters_case_insensitive_matching():
    # the reference hashes only the first four characters of an option, case-insensitively
    assert _params(["-TOPK", "100"]).feature_count_threshold == 100
    assert _params(["-topkselection", "100"]).feature_count_threshold == 100
    assert _params(["-HALFsift"]).half_sift == 1
    assert _params(["-loweorigin"]).lowe_origin == 1
    # unknown options are dropped silently, as -nogl is in the reference (hessgpucmd.cpp:33)
    p = _params(["-nogl", "-zzz", "7", "-t", "0.01"])
    assert abs(p.dog_threshold - 0.01) < 1e-9


def test_m_and_s_read_but_do_not_consume_their_value():
    # "-m 3 -t 0.01": 3 is read as the orientation count, then skipped as a non-option token
    p = _params(["-m", "3", "-t", "0.01"])
    assert p.max_orientation == 3 and abs(p.dog_threshold - 0.01) < 1e-9
    assert _params(["-m"]).max_orientation == 2
    assert _params(["-m", "9"]).max_orientation == 4 and _params(["-m", "0"]).max_orientation == 1
    assert _params(["-s", "0"]).subpixel == 0 and _params(["-s"]).subpixel == 1


def test_image_list_options():
    s = siftgpu_lib.SiftGPU(["-i", "a.pgm", "b.pgm", "c.pgm", "-t", "0.01"])
    assert s.L.siftgpu_image_count(s.h) == 3
    assert abs(s.params().dog_threshold - 0.01) < 1e-9
    s.close()


def test_no_device_no_context():
    import torch

    if torch.cuda.is_available():
        pytest.skip("GPU present")
    s = siftgpu_lib.SiftGPU([])
    assert s.create_context() == 0  # SIFTGPU_NOT_SUPPORTED, no CPU fallback
    import numpy as np
    assert s.run(np.zeros((32, 32), np.uint8), siftgpu_lib.GL_LUMINANCE, siftgpu_lib.GL_UNSIGNED_BYTE) == 0
    s.close()


def test_jpeg_decoding_hook():
    """The file loader of RunSIFT(path) without a device (siftgpu_debug_load_image): data/640-1.jpg through libjpeg looked
    up at run time, against PIL's decode of the same file (decoders differ in chroma up-sampling: sub-level mean difference)."""
    import ctypes as C
    import os

    import numpy as np

    import fixtures
    import siftgpu_lib

    L = siftgpu_lib.lib()
    L.siftgpu_debug_load_image.restype = C.c_int
    L.siftgpu_debug_load_image.argtypes = [C.c_char_p, C.c_void_p, C.c_size_t, C.POINTER(C.c_int), C.POINTER(C.c_int)]
    buf = np.zeros(640 * 480 * 3, dtype=np.uint8)
    w, h = C.c_int(), C.c_int()
    fmt = L.siftgpu_debug_load_image(os.path.join(fixtures._DATA, "640-1.jpg").encode(), buf.ctypes.data, buf.size, C.byref(w), C.byref(h))
    if fmt == -1:
        import pytest
        pytest.skip("no libjpeg of the compiled-in version at run time (or the build found no jpeglib.h)")
    assert fmt == 3 and (w.value, h.value) == (640, 480)          # HESS_FMT_RGB
    ref = fixtures.load_rgb("640-1.jpg").astype(np.int32)
    got = buf.reshape(480, 640, 3).astype(np.int32)
    assert np.abs(got - ref).mean() < 1.5
    # a PNG and a PGM through the same hook
    big = np.zeros(2048 * 2048 * 4, dtype=np.uint8)
    assert L.siftgpu_debug_load_image(os.path.join(fixtures._DATA, "blobs.png").encode(), big.ctypes.data, big.size, C.byref(w), C.byref(h)) in (1, 2, 3, 4)
    # (box.pgm carries a "# CREATOR" comment line, which the reference's PNM loader does not accept either: GLTexImage.cpp:1164)
    assert L.siftgpu_debug_load_image(os.path.join(fixtures._DATA, "box.pgm").encode(), big.ctypes.data, big.size, C.byref(w), C.byref(h)) == 0
    import tempfile
    with tempfile.TemporaryDirectory() as td:
        pgm = os.path.join(td, "t.pgm")
        with open(pgm, "wb") as f:
            f.write(b"P5\n4 2\n255\n" + bytes(range(8)))
        assert L.siftgpu_debug_load_image(pgm.encode(), big.ctypes.data, big.size, C.byref(w), C.byref(h)) == 1
        assert (w.value, h.value) == (4, 2) and list(big[:8]) == list(range(8))
