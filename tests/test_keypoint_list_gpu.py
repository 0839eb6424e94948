"""User-supplied keypoint lists: SiftGPU::SetKeypointList / RunSIFT(num, keys, flag)
-> hess_set_keypoints / hess_run_keypoints, HIP path against the oracle (bit-exact)."""
import numpy as np
import pytest

import fixtures
from hessgpu_amd import _abi
from oracle_lib import OracleSession

pytestmark = pytest.mark.gpu


def _pair(gpu_ctx_factory, **kw):
    return gpu_ctx_factory(**kw), OracleSession(threads=8, **kw)


def _same(g, o, what):
    gk, gd = g.fetch(0)
    ok, od = o.fetch(0)
    assert len(gk) == len(ok) > 0, what
    assert gk.tobytes() == ok.tobytes(), f"{what}: keypoints differ"
    assert np.array_equal(gd.view(np.uint32), od.view(np.uint32)), f"{what}: descriptors differ"


@pytest.mark.parametrize("have_orientation", [1, 0])
@pytest.mark.parametrize("kw", [dict(), dict(max_orientation=1), dict(half_sift=1), dict(fixed_orientation=1)])
def test_run_keypoints_on_current_image(gpu_ctx_factory, kw, have_orientation):
    img = fixtures.load_rgb("640-1.jpg")
    g, o = _pair(gpu_ctx_factory, **kw)
    g.run(img[None]); o.run(img[None])
    keys, _ = o.fetch(0)
    # perturb scales so that some keys change level and the first/last level catch-alls are hit
    keys = keys.copy()
    keys["s"][::7] *= 3.1
    keys["s"][3::11] *= 0.2
    keys["s"][5] = 400.0
    assert g.run_keypoints(keys, have_orientation) == o.run_keypoints(keys, have_orientation) == len(keys)
    _same(g, o, f"run_keypoints {kw} orient={have_orientation}")
    gk, _ = g.fetch(0)
    if have_orientation or not (kw.get("max_orientation") == 1 or kw.get("fixed_orientation")):
        assert gk.tobytes() == keys.tobytes()  # the caller's keypoints come back unchanged
    # the list is consumed: the next plain run detects again
    assert g.run(img[None]) == o.run(img[None])
    _same(g, o, "detection after a keypoint run")


def test_set_keypoints_then_run_on_another_image(gpu_ctx_factory):
    a, b = fixtures.load_rgb("640-2.jpg"), fixtures.load_rgb("640-3.jpg")
    g, o = _pair(gpu_ctx_factory)
    g.run(a[None]); o.run(a[None])
    keys, _ = o.fetch(0)
    keys = keys[:300]
    for s in (g, o):
        s.set_keypoints(keys, have_orientation=False)
    assert g.run(b[None]) == o.run(b[None]) == [len(keys)]
    _same(g, o, "set_keypoints + run")
    # more keypoints than the top-K storage would hold
    g2, o2 = _pair(gpu_ctx_factory, truncate_method=_abi.TRUNC_TOPK, feature_count_threshold=50)
    g2.run(a[None]); o2.run(a[None])
    many = np.concatenate([keys, keys, keys])
    assert g2.run_keypoints(many, 1) == o2.run_keypoints(many, 1) == len(many)
    _same(g2, o2, "list larger than top-K")


def test_keypoints_at_borders_and_corners(gpu_ctx_factory):
    """User keypoints on and beyond the image border: sample boxes clipped to one or two columns/rows
    or to nothing, at small and large scales, with and without given orientations."""
    img = fixtures.load_rgb("sunflowers.png")   # 768 x 323: odd sizes in every octave
    h, w = img.shape[:2]
    pts = []
    for x in (0.0, 0.6, 1.4, 2.5, w / 2.0, w - 2.5, w - 1.4, w - 0.6, float(w)):
        for y in (0.0, 0.7, 1.6, h / 2.0, h - 1.6, h - 0.7, float(h)):
            for sc in (0.8, 1.7, 4.2, 11.0, 37.0):
                pts.append((x, y, sc, (x * 0.37 + y * 0.11 + sc) % 6.28))
    keys = np.zeros(len(pts), dtype=_abi.KEYPOINT_DTYPE)
    keys["x"], keys["y"], keys["s"], keys["o"] = (np.array(v, dtype=np.float32) for v in zip(*pts))
    for have_orientation in (1, 0):
        g, o = _pair(gpu_ctx_factory)
        g.run(img[None]); o.run(img[None])
        assert g.run_keypoints(keys, have_orientation) == o.run_keypoints(keys, have_orientation) == len(keys)
        _same(g, o, f"border keypoints orient={have_orientation}")


def test_dynamic_indexing_option(gpu_ctx_factory):
    """-di (GlobalUtil::_UseDynamicIndexing): a sample whose bin coordinate rounds up to exactly 8.0 goes
    to des[8] instead of being dropped.  A float ramp with a minute vertical slope and keypoints of
    orientation 0 makes (0 - theta)*4/pi + 8 round to 8.0 for most samples, so the two modes differ."""
    h, w = 96, 128
    yy, xx = np.mgrid[0:h, 0:w].astype(np.float32)
    img = (xx / w + 1e-7 * yy / h + 0.05 * np.sin(xx * 0.9) * np.sin(yy * 0.7)).astype(np.float32)
    keys = np.zeros(6, dtype=_abi.KEYPOINT_DTYPE)
    keys["x"] = [30, 50, 64, 80, 100, 64]; keys["y"] = [30, 48, 40, 60, 50, 70]
    keys["s"] = [2.0, 3.0, 2.5, 4.0, 2.2, 6.0]; keys["o"] = 0.0
    out = {}
    for di in (0, 1):
        g, o = _pair(gpu_ctx_factory, dynamic_indexing=di)
        g.run(img[None]); o.run(img[None])
        assert g.run_keypoints(keys, 1) == o.run_keypoints(keys, 1) == len(keys)
        _same(g, o, f"dynamic_indexing={di}")
        out[di] = g.fetch(0)[1].copy()
    # keypoint angle float(2*pi) on ramps whose gradient direction is 0 (or a few 1e-7 rad): the kernel's
    # angle is then a hair below the pixel's, (angle - theta)*4/pi is within half an ulp below 0 and
    # "+ 8" rounds to exactly 8.0 for (nearly) every sample: dropped without -di, added to des[8] with it
    differs = 0
    keys["o"] = np.float32(6.2831855)
    for slope in (0.0, 2.0e-7, 1.0e-6):
        img2 = (xx.astype(np.float64) / w + slope * yy.astype(np.float64) / h).astype(np.float32)
        res = {}
        for di in (0, 1):
            g, o = _pair(gpu_ctx_factory, dynamic_indexing=di, normalize=0)
            g.run(img2[None]); o.run(img2[None])
            g.run_keypoints(keys, 1); o.run_keypoints(keys, 1)
            _same(g, o, f"ramp {slope}, dynamic_indexing={di}")
            res[di] = g.fetch(0)[1].copy()
        differs += int(not np.array_equal(res[0], res[1]))
    out[10], out[11] = (0, 1) if differs == 3 else (0, 0)
    assert not np.array_equal(out[10], out[11]), "the -di path was not exercised"


def test_keypoint_list_errors(gpu_ctx_factory):
    from hessgpu_amd.session import HessError

    g = gpu_ctx_factory()
    keys = np.zeros(4, dtype=_abi.KEYPOINT_DTYPE)
    keys["s"] = 2.0; keys["x"] = 50; keys["y"] = 40
    with pytest.raises(HessError):
        g.run_keypoints(keys, 1)  # no current image
    imgs = np.stack([fixtures.synthetic_blobs(160, 120, i) for i in range(2)])
    g.set_keypoints(keys, True)
    with pytest.raises(HessError):
        g.run(imgs)  # a list applies to one image


def test_siftgpu_class_keypoint_list():
    import siftgpu_lib

    img = fixtures.load_rgb("640-4.jpg")
    s = siftgpu_lib.SiftGPU([])
    assert s.run(img, siftgpu_lib.GL_RGB, siftgpu_lib.GL_UNSIGNED_BYTE) == 1
    k, d = s.features()
    L = s.L
    import ctypes as C
    L.siftgpu_run_keys.restype = C.c_int
    L.siftgpu_run_keys.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int]
    assert L.siftgpu_run_keys(s.h, len(k), k.ctypes.data, 1) == 1
    k2, d2 = s.features()
    o = OracleSession(threads=8)
    o.run(img[None])
    assert o.run_keypoints(k, 1) == len(k)
    ok, od = o.fetch(0)
    assert k2.tobytes() == ok.tobytes() and np.array_equal(d2.view(np.uint32), od.view(np.uint32))
    s.close()
