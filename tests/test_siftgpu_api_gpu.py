"""The SiftGPU C++ class end to end on the GPU: RunSIFT / GetFeatureNum / GetFeatureVector /
SaveSIFT through libsiftgpu.so, compared with the oracle."""
import os
import struct

import numpy as np
import pytest

import fixtures
import siftgpu_lib
from oracle_lib import OracleSession

pytestmark = pytest.mark.gpu


def test_runsift_rgb_matches_oracle():
    img = fixtures.load_rgb("640-1.jpg")
    s = siftgpu_lib.SiftGPU([])
    assert s.create_context() == 2  # SIFTGPU_FULL_SUPPORTED
    assert s.run(img, siftgpu_lib.GL_RGB, siftgpu_lib.GL_UNSIGNED_BYTE) == 1
    k, d = s.features()
    o = OracleSession(threads=8, keep_levels=False)
    o.run(img[None])
    ok, od = o.fetch(0)
    assert k.tobytes() == ok.tobytes() and np.array_equal(d.view(np.uint32), od.view(np.uint32))
    t = s.timing()
    assert t[11] > 0 and t[2] == 0  # TIMINGS_TOTAL; the wrapper passes -v 0: no stage timers (no events between the stages)
    s.close()
    # -v 2 and more (SiftGPU.cpp:433-464: _timingS): stage times in _timing[2..10], same features
    s = siftgpu_lib.SiftGPU(["-v", "2"])
    assert s.create_context() == 2 and s.run(img, siftgpu_lib.GL_RGB, siftgpu_lib.GL_UNSIGNED_BYTE) == 1
    k2, d2 = s.features()
    assert k2.tobytes() == ok.tobytes() and np.array_equal(d2.view(np.uint32), od.view(np.uint32))
    t = s.timing()
    assert t[11] > 0 and t[2] > 0 and t[3] > 0 and t[8] > 0  # total, pyramid, detection, descriptors
    assert abs(sum(t[i] for i in (2, 3, 4, 5, 6, 8, 10)) + t[0] - t[11]) < 0.05 * t[11] + 0.05
    s.close()


def test_parse_param_drives_the_path(tmp_path):
    img = fixtures.load_rgb("640-2.jpg")
    s = siftgpu_lib.SiftGPU(["-topk", "200", "-half", "-t", "0.005"])
    assert s.run(img, siftgpu_lib.GL_RGB, siftgpu_lib.GL_UNSIGNED_BYTE) == 1
    k, d = s.features()
    o = OracleSession(threads=8, keep_levels=False, truncate_method=3, feature_count_threshold=200, half_sift=1,
                      dog_threshold=0.005)
    o.run(img[None])
    ok, od = o.fetch(0)
    assert d.shape[1] == 64 and k.tobytes() == ok.tobytes() and np.array_equal(d.view(np.uint32), od.view(np.uint32))
    # options marked * can change after initialisation
    s.parse(["-topk", "50"])
    assert s.run(img, siftgpu_lib.GL_RGB, siftgpu_lib.GL_UNSIGNED_BYTE) == 1
    assert len(np.unique(np.stack([s.features()[0]["x"], s.features()[0]["y"]]), axis=1).T) == 50
    s.close()


def test_pgm_file_and_save_formats(tmp_path):
    lum = fixtures.load_rgb("640-3.jpg")[..., 1].copy()
    pgm = tmp_path / "img.pgm"
    with open(pgm, "wb") as f:
        f.write(b"P5\n%d %d\n255\n" % (lum.shape[1], lum.shape[0]))
        f.write(lum.tobytes())
    s = siftgpu_lib.SiftGPU([])
    assert s.run_file(str(pgm)) == 1
    k, d = s.features()
    o = OracleSession(threads=8, keep_levels=False)
    o.run(lum[None])
    ok, od = o.fetch(0)
    assert k.tobytes() == ok.tobytes() and np.array_equal(d.view(np.uint32), od.view(np.uint32))
    # text format (SiftPyramid.cpp:504-566)
    txt = tmp_path / "out.sift"
    s.save(str(txt))
    tok = open(txt).read().split()
    n, dim = int(tok[0]), int(tok[1])
    assert n == len(k) and dim == 128
    rec = tok[2:2 + 7 + 128]
    assert abs(float(rec[0]) - k["y"][0]) < 0.006 and abs(float(rec[1]) - k["x"][0]) < 0.006
    assert int(rec[5]) == k["type"][0] and int(rec[6]) == k["level"][0]
    assert [int(v) for v in rec[7:]] == [int(np.floor(0.5 + 512.0 * v)) for v in d[0]]
    # binary format (SiftPyramid.cpp:453-502)
    s.parse(["-b"])
    binp = tmp_path / "out.bin"
    s.save(str(binp))
    raw = open(binp, "rb").read()
    nb, db = struct.unpack("<ii", raw[:8])
    assert (nb, db) == (n, 128) and len(raw) == 8 + n * (24 + 512)
    y, x, sc, ori, resp, typ, lvl = struct.unpack("<fffffHH", raw[8:32])
    assert (x, y, sc, ori, resp, typ, lvl) == tuple(k[f][0] for f in ("x", "y", "s", "o", "response", "type", "level"))
    assert np.array_equal(np.frombuffer(raw[32:32 + 512], np.float32), d[0])
    # vlfeat-style format (SiftPyramid.cpp:372-448)
    s.parse(["-bvlf"])
    vl = tmp_path / "out.vlf"
    s.save(str(vl))
    raw = open(vl, "rb").read()
    assert raw[:4] == b"aff\x01" and struct.unpack("<iiii", raw[4:20]) == (n, 128, 640, 480)
    assert len(raw) == 20 + n * (9 * 4 + 128)
    s.close()


def test_oversize_image_returns_zero_not_exit():
    s = siftgpu_lib.SiftGPU([])
    assert s.run(np.zeros((8, 3300), np.uint8), siftgpu_lib.GL_LUMINANCE, siftgpu_lib.GL_UNSIGNED_BYTE) == 0
    s.parse(["-maxd", "4096"])
    assert s.run(np.zeros((8, 3300), np.uint8), siftgpu_lib.GL_LUMINANCE, siftgpu_lib.GL_UNSIGNED_BYTE) == 1
    s.close()


def test_failed_run_leaves_no_stale_results():
    """Good run, failing run, parameter change, run: the instance never hands out (or copies) the earlier run's
    records after a failure (round-3 advisor finding: the context was rebuilt with 1-element arrays and hess_fetch
    copied the stale count into them)."""
    img = fixtures.load_rgb("640-1.jpg")
    lum = np.ascontiguousarray(img[..., 1])
    s = siftgpu_lib.SiftGPU([])
    assert s.run(lum, siftgpu_lib.GL_LUMINANCE, siftgpu_lib.GL_UNSIGNED_BYTE) == 1
    n_good = s.L.siftgpu_feature_num(s.h)
    assert n_good > 100
    # a run that fails inside the library ("image too small"): no features, nothing to fetch
    assert s.run(np.zeros((2, 2), np.uint8), siftgpu_lib.GL_LUMINANCE, siftgpu_lib.GL_UNSIGNED_BYTE) == 0
    assert s.L.siftgpu_feature_num(s.h) == 0
    k, d = s.features()
    assert len(k) == 0
    # a dirty parameter rebuilds the context on the next run (drop_context -> materialize_results)
    s.L.siftgpu_set_verbose(s.h, 2)
    s.parse(["-maxd", "3000"])
    assert s.run(lum, siftgpu_lib.GL_LUMINANCE, siftgpu_lib.GL_UNSIGNED_BYTE) == 1
    assert s.L.siftgpu_feature_num(s.h) == n_good
    k2, d2 = s.features()
    o = OracleSession(threads=8, keep_levels=False)
    o.run(lum[None])
    ok, od = o.fetch(0)
    assert k2.tobytes() == ok.tobytes() and np.array_equal(d2.view(np.uint32), od.view(np.uint32))
    # and the C ABI itself refuses the stale batch after a failed run
    import hessgpu_amd
    g = hessgpu_amd.HessContext(0)
    assert g.run(lum[None])[0] == n_good
    with pytest.raises(hessgpu_amd.HessError):
        g.run(np.zeros((1, 2, 2), np.uint8))
    with pytest.raises(hessgpu_amd.HessError):
        g.count(0)
    with pytest.raises(hessgpu_amd.HessError):
        g.fetch(0)
    assert g.run(lum[None])[0] == n_good
    g.close()
    s.close()


@pytest.mark.parametrize("name,params", [("sunflowers.png", ["-t", "0.02", "-topk", "10"]), ("blobs.png", []),
                                         ("checkerboard.png", ["-t", "0.000001", "-topk", "500"])])
def test_png_files_are_read_through_libpng_at_run_time(name, params):
    """RunSIFT(path) on the reference's own PNG demo inputs: decoded by libpng16 (dlopen, no build dependency) into the
    file's own layout (RGB; RGBA for the checkerboard's tRNS) and handed to the conversion the reference's DevIL path
    feeds (GLTexImage.cpp:1136-1142 -> SetImageData) -- same features as the oracle on PIL-decoded pixels."""
    s = siftgpu_lib.SiftGPU(["-maxd", "4096"] + params)
    assert s.run_file(os.path.join(fixtures._DATA, name)) == 1
    k, d = s.features()
    kw = {}
    it = iter(params)
    for a in it:
        v = next(it)
        if a == "-t":
            kw["dog_threshold"] = float(v)
        if a == "-topk":
            kw.update(truncate_method=3, feature_count_threshold=int(v))
    o = OracleSession(threads=16, keep_levels=False, tex_max_dim=4096, **kw)
    o.run(fixtures.load_rgb(name)[None])
    ok, od = o.fetch(0)
    assert len(k) == len(ok) > 0 and k.tobytes() == ok.tobytes() and np.array_equal(d.view(np.uint32), od.view(np.uint32))
    s.close()


def test_unreadable_files_return_zero(tmp_path):
    s = siftgpu_lib.SiftGPU([])
    assert s.create_context() == 2
    bad = tmp_path / "broken.png"
    bad.write_bytes(b"\x89PNG\r\n\x1a\n" + b"not a png at all")
    assert s.run_file(str(bad)) == 0 and s.L.siftgpu_feature_num(s.h) == 0
    bad_jpg = tmp_path / "broken.jpg"
    bad_jpg.write_bytes(b"\xff\xd8\xff\xe0" + b"not a jpeg at all" * 10)
    assert s.run_file(str(bad_jpg)) == 0 and s.L.siftgpu_feature_num(s.h) == 0    # libjpeg's error_exit comes back as 0
    assert s.run_file(str(tmp_path / "missing.pgm")) == 0
    s.close()


def test_jpeg_files_are_read_through_libjpeg_at_run_time():
    """RunSIFT(path) on the reference's own JPEG inputs (data/640-1.jpg; the reference decodes them with DevIL,
    GLTexImage.cpp:1117-1158): decoded by libjpeg looked up at run time.  Decoders differ in their chroma up-sampling (this
    image's libjpeg is IJG 9, the tests' PIL carries libjpeg-turbo: 0.5 grey levels apart on average), so the features are
    compared with the oracle's on PIL-decoded pixels by count and position, not bit for bit."""
    s = siftgpu_lib.SiftGPU([])
    assert s.run_file(os.path.join(fixtures._DATA, "640-1.jpg")) == 1
    k, d = s.features()
    o = OracleSession(threads=16, keep_levels=False)
    o.run(fixtures.load_rgb("640-1.jpg")[None])
    ok, od = o.fetch(0)
    assert len(ok) > 150 and abs(len(k) - len(ok)) <= 0.03 * len(ok)
    # most of the oracle's keypoints have a detected twin within a pixel
    import scipy.spatial
    tree = scipy.spatial.cKDTree(np.stack([k["x"], k["y"]], axis=1))
    dist, _ = tree.query(np.stack([ok["x"], ok["y"]], axis=1))
    assert np.mean(dist < 1.0) > 0.9
    s.close()
