"""GPU parity: the HIP path (through the C ABI, libhessgpu.so) against the CPU oracle on the same
inputs.  Bar: bit-exact for every stage -- pyramid planes, det-Hessian, gradient/theta, the raw
detection list (integer positions, packed half response, offsets), keypoints and descriptors --
because both sides evaluate the same IEEE binary32 operation sequences (DESIGN.md "Arithmetic
model").  The north star's tolerance for descriptors/keypoints is 1e-4; the tests assert 0 first
and report the max deviation if that ever fails.
"""
import os

import numpy as np
import pytest

import fixtures
from hessgpu_amd import _abi
from oracle_lib import OracleSession

pytestmark = pytest.mark.gpu

TOL = 1e-4  # BASELINE.json north_star: "keypoints/descriptors matching reference within 1e-4"
# The product's default descriptor order (HESS_DESC_ORDER_PIXEL) is not the reference's summation order (include/hess_abi.h,
# hess_params.descriptor_order).  On every BASELINE config it is tied to the reference two ways:
#   TOL_EXACT   to the reference's FORMULA (ProgramCU.cu:1690-1790 + :1950-2054) evaluated in double precision by the
#               oracle (oracle/hess_oracle.h, HESS_ORACLE_DESC_EXACT) -- measured <= 2.5e-7 on configs[1], [2], [4];
#   TOL_ORDER   to the reference's sequential FLOAT order (the oracle's descriptor_order=1).  That order carries its own
#               rounding -- it rounds every cell centre at the magnitude of the image coordinate, 1e-5 relative at
#               x = 4000 -- and is itself 6e-6 (1080p), 2.5e-6 (640x480), 1.6e-5 (4096^2) from the formula, so the
#               distance between the two orders is that figure; the test bounds it by the sequential order's own
#               distance to the formula + TOL_EXACT, and by TOL_ORDER overall (north star: 1e-4).
TOL_EXACT = 1e-6
TOL_ORDER = 3e-5
HESS_ORACLE_DESC_EXACT = 3


def _assert_tied_to_reference_order(g, imgs, kw, what, threads=16):
    """The context's results (already run on imgs, any descriptor order) against the oracle in the REFERENCE's
    summation order and against the reference's formula in double precision: keypoints bitwise, descriptors within
    the tolerances above."""
    worst = {"seq": 0.0, "exact": 0.0, "seq_exact": 0.0}
    os_ = OracleSession(threads=threads, keep_levels=False, descriptor_order=_abi.DESC_ORDER_SEQUENTIAL, **kw)
    oe = OracleSession(threads=threads, keep_levels=False, descriptor_order=HESS_ORACLE_DESC_EXACT, **kw)
    os_.run(imgs)
    oe.run(imgs)
    for b in range(len(imgs)):
        gk, gd = g.fetch(b)
        sk, sd = os_.fetch(b)
        ek, ed = oe.fetch(b)
        assert gk.tobytes() == sk.tobytes() == ek.tobytes(), f"{what}: img {b}: keypoints differ from the sequential-order oracle"
        if sd.size:
            ed = ed.astype(np.float64)
            worst["seq"] = max(worst["seq"], float(np.abs(gd - sd.astype(np.float64)).max()))
            worst["exact"] = max(worst["exact"], float(np.abs(gd - ed).max()))
            worst["seq_exact"] = max(worst["seq_exact"], float(np.abs(sd - ed).max()))
    os_.close()
    oe.close()
    if g.params.descriptor_order == _abi.DESC_ORDER_PIXEL:
        assert worst["exact"] <= TOL_EXACT, f"{what}: distance to the reference's formula in double precision {worst}"
        assert worst["seq"] <= min(TOL_ORDER, worst["seq_exact"] + TOL_EXACT), f"{what}: distance to the sequential float order {worst}"
    else:  # a float order: within rounding of the sequential one (tests/test_descriptor_order.py: 1e-6), as far from the formula as it
        assert worst["seq"] <= 1e-6 and worst["exact"] <= worst["seq_exact"] + 1e-6, f"{what}: {worst}"
    return worst


def _assert_same_features(gk, gd, ok, od, what):
    assert len(gk) == len(ok), f"{what}: feature count {len(gk)} != oracle {len(ok)}"
    assert np.array_equal(gk["level"], ok["level"]) and np.array_equal(gk["type"], ok["type"]), f"{what}: level/type"
    for f in ("x", "y", "s", "o", "response"):
        if not np.array_equal(gk[f], ok[f]):
            d = np.max(np.abs(gk[f].astype(np.float64) - ok[f].astype(np.float64)))
            assert d <= TOL, f"{what}: keypoint field {f} max abs diff {d}"
            pytest.fail(f"{what}: keypoint field {f} within 1e-4 (max {d}) but not bit-exact")
    if od.size:
        if not np.array_equal(gd.view(np.uint32), od.view(np.uint32)):
            d = np.nanmax(np.abs(gd.astype(np.float64) - od.astype(np.float64)))
            bad = np.sum(np.any(gd.view(np.uint32) != od.view(np.uint32), axis=1))
            assert d <= TOL, f"{what}: descriptors max abs diff {d} ({bad} rows differ)"
            pytest.fail(f"{what}: descriptors within 1e-4 (max {d}, {bad} rows) but not bit-exact")


def _compare_all(g, o, imgs, what, stages=True):
    g.keep_levels(stages)   # the top Gaussian level of an octave is only written to HBM on request (hess_debug_keep_levels)
    ng = g.run(imgs)
    no = o.run(imgs)
    assert g.geometry() == o.geometry()
    if stages:
        nlev = o.params.dog_level_num + 2
        for b in range(len(no)):
            for oc in range(len(o.geometry())):
                for l in range(nlev):
                    a, r = g.level(b, oc, l, _abi.DBG_GAUSS), o.level(b, oc, l, _abi.DBG_GAUSS)
                    assert np.array_equal(a.view(np.uint32), r.view(np.uint32)), \
                        f"{what}: gauss img {b} oct {oc} lvl {l}: {np.sum(a != r)} px differ, max {np.max(np.abs(a - r))}"
                for l in range(nlev):
                    a, r = g.level(b, oc, l, _abi.DBG_DETH), o.level(b, oc, l, _abi.DBG_DETH)
                    assert np.array_equal(a.view(np.uint32), r.view(np.uint32)), \
                        f"{what}: det-H img {b} oct {oc} lvl {l}: {np.sum(a != r)} px differ, max {np.max(np.abs(a - r))}"
                for l in range(1, nlev - 1):
                    a, r = g.level(b, oc, l, _abi.DBG_GOT), o.level(b, oc, l, _abi.DBG_GOT)
                    assert np.array_equal(a.view(np.uint32), r.view(np.uint32)), \
                        f"{what}: grad/theta img {b} oct {oc} lvl {l}: {np.sum(a != r)} values differ"
    for b in range(len(no)):
        gl, ol = g.rawlist(b), o.rawlist(b)
        assert len(gl) == len(ol), f"{what}: img {b} list length {len(gl)} != {len(ol)}"
        assert gl.tobytes() == ol.tobytes(), f"{what}: img {b} detection list differs"
    assert ng == no, f"{what}: feature counts {ng} != {no}"
    for b in range(len(no)):
        gk, gd = g.fetch(b)
        ok, od = o.fetch(b)
        _assert_same_features(gk, gd, ok, od, f"{what} img {b}")
    return no


def test_math_functions_bit_exact(gpu_ctx_factory):
    import ctypes as C
    from oracle_lib import lib as olib

    g = gpu_ctx_factory()
    L = olib()
    rng = np.random.RandomState(7)
    x = np.concatenate([-rng.rand(20000) * 90.0, [0.0, -87.0, -87.5, -1e-8]]).astype(np.float32)
    got = g.math_probe(0, x)
    ref = np.array([L.hess_cpu_expf(float(v)) for v in x], dtype=np.float32)
    assert np.array_equal(got.view(np.uint32), ref.view(np.uint32))
    a = (rng.randn(20000) * 0.3).astype(np.float32)
    b = (rng.randn(20000) * 0.3).astype(np.float32)
    a[:4] = [0, 0, 1, -1]; b[:4] = [0, 1, 0, 0]
    got = g.math_probe(1, a, b)
    ref = np.array([L.hess_cpu_atan2f(float(p), float(q)) for p, q in zip(a, b)], dtype=np.float32)
    assert np.array_equal(got.view(np.uint32), ref.view(np.uint32))
    ang = (rng.rand(20000) * 2 * np.pi).astype(np.float32)
    s, c = g.math_probe(2, ang), g.math_probe(3, ang)
    rs, rc = np.zeros_like(ang), np.zeros_like(ang)
    fs, fc = C.c_float(), C.c_float()
    for i, v in enumerate(ang):
        L.hess_cpu_sincosf(float(v), C.byref(fs), C.byref(fc))
        rs[i], rc[i] = fs.value, fc.value
    assert np.array_equal(s.view(np.uint32), rs.view(np.uint32)) and np.array_equal(c.view(np.uint32), rc.view(np.uint32))
    v = np.concatenate([rng.randn(20000) * 0.05, rng.randn(2000) * 1e-6, [0, 65504, 65519.9, 65520, 1e-8, 6e-8]]).astype(np.float32)
    h = g.math_probe(4, v)
    assert np.array_equal(h.astype(np.uint16), v.astype(np.float16).view(np.uint16))
    # IEEE division and square root on the device (u8/255 conversion, Gaussian elimination, gradient)
    num, den = rng.rand(20000).astype(np.float32), (rng.rand(20000) + 0.1).astype(np.float32)
    assert np.array_equal(g.math_probe(6, num, den), num / den)
    assert np.array_equal(g.math_probe(7, num), np.sqrt(num))
    # u8 luminance -> [0,1]: the three-instruction form equals the IEEE quotient p / 255.0f for every byte value
    byte = np.arange(256, dtype=np.float32)
    assert np.array_equal(g.math_probe(8, byte).view(np.uint32), (byte / np.float32(255.0)).view(np.uint32))


def test_parity_640_rgb_all_stages(gpu_ctx_factory):
    img = fixtures.load_rgb("640-1.jpg")
    g = gpu_ctx_factory()
    o = OracleSession(threads=8)
    n = _compare_all(g, o, img[None], "640-1.jpg defaults")
    assert n[0] > 100


def test_parity_list640_sequence(gpu_ctx_factory):
    """BASELINE.json configs[2]: data/list640.txt processed on one instance, descriptors checked."""
    g = gpu_ctx_factory()
    o = OracleSession(threads=8, keep_levels=False)
    for name in fixtures.list640():
        img = fixtures.load_rgb(name)
        _compare_all(g, o, img[None], name, stages=False)
        _assert_tied_to_reference_order(g, img[None], {}, name, threads=8)


def test_parity_batch_of_images(gpu_ctx_factory):
    """The five 640x480 images as ONE batch (batch dimension of every kernel)."""
    imgs = np.stack([fixtures.load_rgb(n) for n in fixtures.list640()])
    g = gpu_ctx_factory()
    o = OracleSession(threads=8, keep_levels=False)
    _compare_all(g, o, imgs, "batch of 5", stages=False)


@pytest.mark.parametrize("name", ["blobs.png", "sunflowers.png"])
def test_parity_odd_sizes(gpu_ctx_factory, name):
    """500x500 and 768x323: widths that are re-aligned per octave, padded columns scanned."""
    img = fixtures.load_rgb(name)
    g = gpu_ctx_factory()
    o = OracleSession(threads=8)
    _compare_all(g, o, img[None], name)


def test_parity_1080p_topk(gpu_ctx_factory):
    """BASELINE.json configs[1]: 1920x1080 synthetic blobs, default octaves/levels, top-K=4096."""
    im = fixtures.synthetic_blobs(1920, 1080, 0)
    kw = dict(truncate_method=_abi.TRUNC_TOPK, feature_count_threshold=4096)
    g = gpu_ctx_factory(**kw)
    o = OracleSession(threads=8, **kw)
    n = _compare_all(g, o, im[None], "1080p top-K 4096")
    assert len(o.rawlist(0)) == 4096 and n[0] >= 4096
    _assert_tied_to_reference_order(g, im[None], kw, "1080p top-K 4096")


def test_parity_4096_half_topk65536(gpu_ctx_factory):
    """BASELINE.json configs[4]: 4096x4096 single image, top-K=65536, <0,PI> unsigned-gradient (half)
    descriptor mode; needs -maxd 4096 (tex_max_dim) exactly as the reference would."""
    im = fixtures.synthetic_blobs(4096, 4096, 5)
    kw = dict(truncate_method=_abi.TRUNC_TOPK, feature_count_threshold=65536, half_sift=1, tex_max_dim=4096)
    g = gpu_ctx_factory(**kw)
    o = OracleSession(threads=16, keep_levels=False, **kw)
    n = _compare_all(g, o, im[None], "4096x4096 half top-K 65536", stages=False)
    assert len(o.geometry()) == 9 and len(o.rawlist(0)) == 65536 and n[0] >= 65536
    _assert_tied_to_reference_order(g, im[None], kw, "4096x4096 half top-K 65536")
    k, d = g.fetch(0)
    assert d.shape[1] == 64
    # A single image whose results are large (28 MB here, judged by the context's batch before), SUBMITTED asynchronously, is
    # delivered by the copier thread in four parts -- four descriptor launches over quarters of the feature list: same bytes
    # as the in-kernel mirror of the context's synchronous runs (compared with the oracle above).
    g.profile_enable(True)
    g.profile_reset()
    g.submit_host(im[None])
    g.wait()
    launches = g.profile()["descriptor"]["launches"]
    g.profile_enable(False)
    k2, d2 = g.fetch(0)
    if not any(v in os.environ for v in ("HESS_DELIVERY", "HESS_DESC_PARTS", "HESS_MIRROR_MAX_MB", "HESS_MIRROR_MAX_BATCH")):   # (the A/B switches choose otherwise)
        assert launches == 4, launches
    assert k2.tobytes() == k.tobytes() and d2.tobytes() == d.tobytes()
    # size-independent properties at full size: unit-norm descriptors, clamp, list order, top-K cut
    assert np.allclose(np.linalg.norm(d, axis=1), 1.0, atol=1e-5) and d.min() >= 0.0
    raw = g.rawlist(0)
    order = np.lexsort((raw["col"], raw["row"], raw["level_index"]))
    assert np.array_equal(order, np.arange(len(raw)))          # (level,row,col) ascending
    resp = np.abs((raw["packed"] >> 16).astype(np.uint16).view(np.float16).astype(np.float32))
    g2 = gpu_ctx_factory(tex_max_dim=4096, compute_descriptors=0, max_orientation=1)
    g2.run(im[None])
    allraw = g2.rawlist(0)
    allresp = np.sort(np.abs((allraw["packed"] >> 16).astype(np.uint16).view(np.float16).astype(np.float32)))[::-1]
    assert len(allraw) > 65536 and resp.min() >= allresp[65535]  # nothing kept is weaker than the K-th


@pytest.mark.parametrize("kw", [
    dict(half_sift=1),
    dict(max_orientation=1),
    dict(fixed_orientation=1),
    dict(subpixel=0),
    dict(compute_descriptors=0),
    dict(lowe_origin=1, dog_threshold=0.004, edge_threshold=5.0),
    dict(dog_level_num=4),
    dict(dog_level_num=1),
    dict(dog_level_num=2),
    dict(dog_level_num=5),
    dict(dog_level_num=6),   # more than 5 levels: LDS-tiled extrema scan instead of the streaming one
    dict(first_octave=1),
    # an up-sampled first octave above -maxd with -ads: the reference raises _octave_min step by step, whatever its
    # sign (PyramidCU.cpp:154-166) -- less up-sampling (-2 -> -1), none (-1 -> 0), and -5 is clamped to -3 first
    dict(first_octave=-2, auto_downscale=1, tex_max_dim=1400),
    dict(first_octave=-1, auto_downscale=1, tex_max_dim=700),
    dict(first_octave=-5, auto_downscale=1, tex_max_dim=1400),
    dict(octave_num=2),
    dict(truncate_method=_abi.TRUNC_TOPK, feature_count_threshold=100),
    dict(truncate_method=_abi.TRUNC_TOPK, feature_count_threshold=100000),
    dict(truncate_method=_abi.TRUNC_HIGHEST_0, feature_count_threshold=150),
    dict(truncate_method=_abi.TRUNC_HIGHEST_1, feature_count_threshold=150),
    dict(truncate_method=_abi.TRUNC_LOWEST, feature_count_threshold=150),
    dict(filter_width_factor=5.0, orient_window_factor=1.5, desc_window_factor=2.5),
])
def test_parity_parameter_variants(gpu_ctx_factory, kw):
    img = fixtures.load_rgb("640-2.jpg")
    g = gpu_ctx_factory(**kw)
    o = OracleSession(threads=8, keep_levels=False, **kw)
    _compare_all(g, o, img[None], str(kw), stages=False)


@pytest.mark.parametrize("dtype,fmt", [("u8lum", _abi.FMT_LUM), ("u16rgb", _abi.FMT_RGB), ("f32lum", _abi.FMT_LUM),
                                       ("f32bgr", _abi.FMT_BGR), ("u8rgba", _abi.FMT_RGBA)])
def test_parity_input_formats(gpu_ctx_factory, dtype, fmt):
    rgb = fixtures.load_rgb("640-3.jpg")
    if dtype == "u8lum":
        img = rgb[..., 1].copy()
    elif dtype == "u16rgb":
        img = rgb.astype(np.uint16) * 257
    elif dtype == "f32lum":
        img = (rgb[..., 1] / 255.0).astype(np.float32)
    elif dtype == "f32bgr":
        img = (rgb[..., ::-1] / 255.0).astype(np.float32)
    else:
        img = np.concatenate([rgb, np.full(rgb.shape[:2] + (1,), 255, np.uint8)], axis=2)
    g = gpu_ctx_factory()
    o = OracleSession(threads=8, keep_levels=False)
    ng = g.run(img[None], fmt=fmt)
    no = o.run(img[None], fmt=fmt)
    assert ng == no and no[0] > 50
    gk, gd = g.fetch(0)
    ok, od = o.fetch(0)
    _assert_same_features(gk, gd, ok, od, dtype)


@pytest.mark.parametrize("w,h", [(251, 50), (124, 24), (125, 25), (372, 73), (1000, 97), (324, 223), (652, 210),
                                 (128, 24), (132, 25), (256, 48), (260, 49), (516, 27)])
def test_parity_noise_ragged(gpu_ctx_factory, w, h):
    """Dense extrema on sizes that straddle the extrema scan's strip (128 columns; 124 until round 3) and segment
    (24 rows) boundaries; a low threshold keeps the candidate queues of the scan full.  The last two sizes walk the
    octave widths 324, 164, 84, 40 and 652, 328, 164, 84: a next octave wider than half of the previous one (width
    aligned up: its last columns repeat the source's last column) and one narrower (84 -> 40: widths halve
    unaligned), the two cases of the decimation fused into the Gaussian launch."""
    rng = np.random.RandomState(w * 131 + h)
    img = (rng.rand(2, h, w) * 255).astype(np.uint8)
    kw = dict(dog_threshold=0.0005, edge_threshold=50.0)
    g = gpu_ctx_factory(dev_switches=True, **kw)
    o = OracleSession(threads=8, **kw)
    _compare_all(g, o, img, f"noise {w}x{h}")


@pytest.mark.parametrize("band", [64, 200, 1000])
def test_descriptor_raster_bands(gpu_ctx_factory, monkeypatch, band):
    """descriptor_pixel_kernel rasters a footprint over the row spans of the rotated window, in bands of <= 64 rows and
    <= 4096 pixels (k_feature.hip).  No detected feature is larger than two bands of rows, so HESS_PX_BAND (developer
    build) shrinks the band: 64 pixels = one row per band and, for boxes wider than 64, bands of columns too; 200 and
    1000 = a few rows per band.  The fixed-point sums do not depend on the order: the bits stay the oracle's -- for
    footprints at every angle (photograph), clipped by the image's border (small image) and for -half."""
    monkeypatch.setenv("HESS_PX_BAND", str(band))
    g = gpu_ctx_factory(dev_switches=True)
    gh = gpu_ctx_factory(dev_switches=True, half_sift=1, first_octave=-1)
    monkeypatch.delenv("HESS_PX_BAND")
    o = OracleSession(threads=8, keep_levels=False)
    oh = OracleSession(threads=8, keep_levels=False, half_sift=1, first_octave=-1)
    img = fixtures.load_rgb("640-1.jpg")
    n = _compare_all(g, o, img[None], f"raster bands of {band} pixels", stages=False)
    assert n[0] > 100
    small = fixtures.synthetic_blobs(96, 80, 5)
    _compare_all(gh, oh, small[None], f"raster bands of {band} pixels, upsampled small image, -half", stages=False)


@pytest.mark.parametrize("mode", ["mirror", "dma", "blit"])
def test_result_delivery_modes(gpu_ctx_factory, monkeypatch, mode):
    """Three ways for the results to reach the host (hess_copier.hip, kDeliver*): stores of the descriptor kernel
    into pinned memory (small batches), a DMA copy issued by the context's copier thread once the kernels are done
    (larger batches), and a device->host copy on the context's stream after hess_wait has read the counts (the
    fallback).  Same results every way, for one image and for a batch, also when modes alternate on one context."""
    monkeypatch.setenv("HESS_DELIVERY", mode)
    g = gpu_ctx_factory()
    monkeypatch.delenv("HESS_DELIVERY")
    imgs = np.stack([fixtures.load_rgb(n) for n in fixtures.list640()[:3]])
    o = OracleSession(threads=8, keep_levels=False)
    _compare_all(g, o, imgs, f"{mode} delivery, batch of 3", stages=False)
    _compare_all(g, o, imgs[:1], f"{mode} delivery, one image", stages=False)
    _compare_all(g, o, imgs[1:], f"{mode} delivery, batch of 2", stages=False)


@pytest.mark.parametrize("parts", [None, "1", "2", "3"])
def test_delivery_in_parts_of_larger_batches(gpu_ctx_factory, monkeypatch, parts):
    """Batches of four or more images delivered by the copier thread get their descriptors in several launches
    (groups of images), a group's results crossing the host link while the next group is computed
    (hess_schedule.hip, enqueue(); HESS_DESC_PARTS=n overrides the number of groups, 1 = one launch, one transfer).
    Same results, also when a group -- or a single image -- has no features at all, and for an odd batch."""
    if parts:
        monkeypatch.setenv("HESS_DESC_PARTS", parts)
    g = gpu_ctx_factory(dev_switches=bool(parts), truncate_method=_abi.TRUNC_TOPK, feature_count_threshold=300)
    o = OracleSession(threads=8, keep_levels=False, truncate_method=_abi.TRUNC_TOPK, feature_count_threshold=300)
    blobs = [fixtures.synthetic_blobs(320, 240, 70 + i) for i in range(5)]
    flat = np.full((240, 320), 128, np.uint8)
    _compare_all(g, o, np.stack(blobs), "batch of 5", stages=False)
    _compare_all(g, o, np.stack(blobs + blobs[:3]), "batch of 8", stages=False)
    n = _compare_all(g, o, np.stack([flat, flat, blobs[0], blobs[1]]), "first half without features", stages=False)
    assert n[0] == 0 and n[1] == 0 and n[2] > 0
    n = _compare_all(g, o, np.stack([blobs[2], blobs[3], flat, flat, flat]), "second half without features", stages=False)
    assert n[1] > 0 and n[2] == 0 and n[4] == 0
    n = _compare_all(g, o, np.stack([flat] * 4), "no features at all", stages=False)
    assert sum(n) == 0
    _compare_all(g, o, np.stack(blobs[:3]), "back to a batch of 3 (one launch)", stages=False)


@pytest.mark.parametrize("side", [True, False])
def test_pinned_input_is_uploaded_beside_the_queues(gpu_ctx_factory, monkeypatch, side):
    """hess_submit_host with pinned pixels and a batch the copier thread delivers: the upload is an SDMA copy started by
    the caller, the copier thread waits for it on the host and enqueues the kernels (no transfer command in the
    context's hardware queue; HESS_NO_SIDE_UPLOAD=1: a copy on the stream as before).  Same results as the oracle
    and as pageable input, also after an overflow that re-runs the batch, for several batches in a row and for a
    batch of two (in-kernel delivery: upload on the stream)."""
    import torch
    if not side:
        monkeypatch.setenv("HESS_NO_SIDE_UPLOAD", "1")
    monkeypatch.setenv("HESS_INITIAL_CAP", "64")     # the first batches overflow and are run again from the device copy
    g = gpu_ctx_factory(dev_switches=True, truncate_method=_abi.TRUNC_TOPK, feature_count_threshold=400)
    monkeypatch.delenv("HESS_INITIAL_CAP")
    o = OracleSession(threads=8, keep_levels=False, truncate_method=_abi.TRUNC_TOPK, feature_count_threshold=400)
    for nimg, seed in ((5, 0), (8, 20), (2, 40), (5, 60)):
        imgs = np.stack([fixtures.synthetic_blobs(320, 240, seed + i) for i in range(nimg)])
        pinned = torch.from_numpy(imgs).pin_memory()
        g.submit_host(ptr=pinned.data_ptr(), batch=nimg, height=240, width=320)
        g.wait()
        want = o.run(imgs)
        assert [g.count(b) for b in range(nimg)] == want and sum(want) > 0
        for b in range(nimg):
            gk, gd = g.fetch(b)
            ok, od = o.fetch(b)
            assert gk.tobytes() == ok.tobytes() and np.array_equal(gd.view(np.uint32), od.view(np.uint32)), (nimg, b)
    assert g.regrown() >= 1


def test_result_delivery_default_switches_with_batch_size(gpu_ctx_factory):
    """Default policy: batches of one or two images use the in-kernel mirror, larger ones the copier thread; a context
    that alternates between them (and pipelines submit/wait pairs) keeps delivering the oracle's results."""
    g = gpu_ctx_factory()
    imgs = np.stack([fixtures.load_rgb(n) for n in fixtures.list640()[:4]])
    o = OracleSession(threads=8, keep_levels=False)
    for sel in ([0, 1, 2, 3], [2], [0, 1, 2], [3, 1], [0, 1, 2, 3]):
        _compare_all(g, o, imgs[sel], f"default delivery, images {sel}", stages=False)
    # submit/wait with the copier: results of the batch submitted last
    g.submit_host(imgs)
    g.wait()
    o.run(imgs)
    for b in range(4):
        gk, gd = g.fetch(b)
        ok, od = o.fetch(b)
        _assert_same_features(gk, gd, ok, od, f"submit/wait img {b}")


@pytest.mark.parametrize("mode", ["mirror", "dma", "blit"])
def test_feature_storage_grows_on_overflow(gpu_ctx_factory, monkeypatch, mode):
    """The reference grows its per-level lists on demand (SetLevelFeatureNum, PyramidCU.cpp:393-397); here a batch
    that overflows the raw-detection or the feature storage raises a device flag, the storage grows and the batch
    runs again (hess_copier.hip, wait_impl).  HESS_INITIAL_CAP=16 makes a new context start with room for 16
    detections per image, so that dense noise at a low threshold overflows both lists -- with every delivery mode,
    for one image and for a batch, with and without top-K."""
    monkeypatch.setenv("HESS_INITIAL_CAP", "16")
    monkeypatch.setenv("HESS_DELIVERY", mode)
    rng = np.random.RandomState(11)
    imgs = (rng.rand(3, 120, 200) * 255).astype(np.uint8)
    kw = dict(dog_threshold=0.0005, edge_threshold=50.0)
    g = gpu_ctx_factory(dev_switches=True, **kw)
    o = OracleSession(threads=8, keep_levels=False, **kw)
    n = _compare_all(g, o, imgs, f"grown storage ({mode})", stages=False)
    assert min(n) > 16 * 4 and g.regrown() >= 1          # both lists had to grow (orientations included)
    grown = g.regrown()
    _compare_all(g, o, imgs, f"grown storage ({mode}), again", stages=False)
    assert g.regrown() == grown                           # grow-only: the second run fits
    kt = dict(kw, truncate_method=_abi.TRUNC_TOPK, feature_count_threshold=30)
    gt = gpu_ctx_factory(dev_switches=True, **kt)
    ot = OracleSession(threads=8, keep_levels=False, **kt)
    _compare_all(gt, ot, imgs[:1], f"grown storage ({mode}), top-K", stages=False)
    assert gt.regrown() >= 1


def test_refused_reservation_leaves_the_context_usable(gpu_ctx_factory):
    """hess_reserve of more memory than the device has returns HESS_ERR_NOMEM (no abort, no exception across the C
    ABI) and the context still runs afterwards, with the right results."""
    from hessgpu_amd.session import HessError
    g = gpu_ctx_factory()
    o = OracleSession(threads=8, keep_levels=False)
    img = fixtures.load_rgb("640-2.jpg")
    _compare_all(g, o, img[None], "before the refused reservation", stages=False)
    with pytest.raises(HessError) as e:
        g.reserve(1920, 1080, 200000)                     # 2.2e13 bytes of planes
    assert e.value.code == _abi.HESS_ERR_NOMEM
    _compare_all(g, o, img[None], "after the refused reservation", stages=False)
    g.reserve(1920, 1080, 2)
    _compare_all(g, o, img[None], "after a reservation that fits", stages=False)


def test_pageable_batch_of_1080p_images_through_the_helper_threads(gpu_ctx_factory):
    """hess_run_host from pageable memory stages inputs of 16 MB and more with four threads (hess_submit_host); a
    batch of eight 1080p images (16.6 MB) takes that path.  Image 0 and 7 against the oracle, and every image
    against the same pixels run one at a time."""
    imgs = np.stack([fixtures.synthetic_blobs(1920, 1080, i) for i in range(2)] * 4)   # pageable numpy memory
    imgs[7] = imgs[7][::-1]                                                           # make the last one distinct
    kw = dict(truncate_method=_abi.TRUNC_TOPK, feature_count_threshold=4096)
    g = gpu_ctx_factory(**kw)
    counts = g.run(imgs)
    res = [g.fetch(b) for b in range(8)]
    o = OracleSession(threads=8, keep_levels=False, **kw)
    for b in (0, 7):
        o.run(imgs[b:b + 1])
        ok, od = o.fetch(0)
        _assert_same_features(res[b][0], res[b][1], ok, od, f"pageable batch img {b}")
    g1 = gpu_ctx_factory(**kw)
    for b in range(8):
        assert g1.run(imgs[b:b + 1]) == [counts[b]]
        k, d = g1.fetch(0)
        assert k.tobytes() == res[b][0].tobytes() and d.tobytes() == res[b][1].tobytes()


def test_edge_cases(gpu_ctx_factory):
    g = gpu_ctx_factory()
    o = OracleSession(threads=1)
    # constant image: no detections, empty outputs
    flat = np.full((1, 64, 64), 128, np.uint8)
    assert g.run(flat) == [0] and o.run(flat) == [0]
    k, d = g.fetch(0)
    assert len(k) == 0 and d.shape == (0, 128)
    # smallest supported image and a width that is not a multiple of 4 (columns dropped)
    rng = np.random.RandomState(3)
    tiny = (rng.rand(1, 17, 23) * 255).astype(np.uint8)
    assert g.run(tiny) == o.run(tiny)
    # image larger than -maxd without -ads: error code, not exit()
    big = np.zeros((1, 8, 3204), np.uint8)
    from hessgpu_amd.session import HessError
    with pytest.raises(HessError) as e:
        g.run(big)
    assert e.value.code == _abi.HESS_ERR_TOO_BIG
    with pytest.raises(HessError):
        o.run(big)
    # pyramid reuse after an error and after a size change
    img = fixtures.load_rgb("640-4.jpg")
    assert g.run(img[None]) == o.run(img[None])


def test_repeatability(gpu_ctx_factory):
    """speed.cpp:121,149 checks only run-to-run feature count; here the full output must repeat."""
    img = fixtures.load_rgb("640-5.jpg")
    g = gpu_ctx_factory()
    g.run(img[None])
    k0, d0 = g.fetch(0)
    for _ in range(3):
        g.run(img[None])
        k, d = g.fetch(0)
        assert k.tobytes() == k0.tobytes() and d.tobytes() == d0.tobytes()


@pytest.mark.gpu
def test_profile_bytes_are_the_layouts_of_survey_8d(gpu_ctx_factory):
    """The roofline numerator of the Gaussian + det-H stage: bytes the launches move (hess_profile_get) + bytes of the arrays
    kept in LDS (hess_profile_get_in_lds) = SURVEY 8(d)'s layout -- per octave pixel Gaussian 5 levels x (4 W + 4 R), det-H
    5 x 4 W, gradient/theta 3 x 8 W = 84 B (level 0 of an octave > 0 is written by the octave before), + 1 B per u8 pixel."""
    B, H, W = 3, 270, 480
    imgs = np.stack([fixtures.synthetic_blobs(W, H, i) for i in range(B)])
    g = gpu_ctx_factory(truncate_method=_abi.TRUNC_TOPK, feature_count_threshold=500)
    g.run(imgs)
    g.profile_enable(True)
    g.profile_reset()
    g.run(imgs)
    prof = g.profile()
    p, ph = prof["gauss"], prof["hessian"]   # (hessian: the top levels' det-H launch of the unfused form, HESS_NO_TOP_FUSION)
    g.profile_enable(False)
    planes = [w * h for w, h in g.geometry()]
    layout = B * (sum(84.0 * px for px in planes) + 1.0 * planes[0])
    assert p["launches"] > 0
    assert abs(p["bytes"] + p["bytes_in_lds"] + ph["bytes"] - layout) < 1e-6 * layout, (p, ph, layout)
    # what stays in LDS: the top level of every octave and level 0 of octave 0, written once and read once in the layout
    top = 0 if "HESS_NO_TOP_FUSION" in os.environ else sum(planes)
    first = 0 if "HESS_NO_FIRST_FUSION" in os.environ else planes[0]
    assert abs(p["bytes_in_lds"] - B * 8.0 * (top + first)) < 1e-6 * layout


def test_a_submitted_pair_runs_like_a_larger_batch_with_the_same_results(gpu_ctx_factory):
    """Two images handed over by hess_submit_* (a pipelining caller) take the level-by-level launches and the copier's
    delivery, two images handed over by hess_run_* the chain launches and the in-kernel mirror: same bytes."""
    imgs = np.stack([fixtures.synthetic_blobs(640, 480, i) for i in range(2)])
    g = gpu_ctx_factory(truncate_method=_abi.TRUNC_TOPK, feature_count_threshold=1000)
    g.run(imgs)
    ref = [g.fetch(b) for b in range(2)]
    for _ in range(2):
        g.submit_host(imgs)
        g.wait()
        for b in range(2):
            k, d = g.fetch(b)
            assert k.tobytes() == ref[b][0].tobytes() and d.tobytes() == ref[b][1].tobytes()
    o = OracleSession(threads=8, keep_levels=False, truncate_method=_abi.TRUNC_TOPK, feature_count_threshold=1000)
    o.run(imgs)
    for b in range(2):
        ok, od = o.fetch(b)
        _assert_same_features(ref[b][0], ref[b][1], ok, od, f"pair img {b}")
    o.close()


def _dot_grid(w, h, seed, period=6, sigma=1.7):
    """Alternating bright / dark Gaussian dots on a grid of `period` pixels: a blob extremum per dot and saddles between
    them -- the densest field of detections the detector admits (about one detection per 20 pixels of a level)."""
    rng = np.random.RandomState(seed)
    y, x = np.mgrid[0:h, 0:w].astype(np.float64)
    img = np.full((h, w), 0.5)
    for cy in range(3, h, period):
        for cx in range(3, w, period):
            a = (0.3 + 0.05 * rng.rand()) * (1 if ((cx // period + cy // period) & 1) else -1)
            img += a * np.exp(-((x - cx) ** 2 + (y - cy) ** 2) / (2 * sigma * sigma))
    return np.clip(img * 255 + rng.rand(h, w) - 0.5, 0, 255).astype(np.uint8)


def test_detection_store_spills_beyond_a_scan_tasks_slots(gpu_ctx_factory):
    """The extrema scan writes every detection into the 64 slots its scan task owns (a wavefront's strip of 124 columns x a
    segment of rows, all detection levels) and what does not fit into the image's spill list (k_detect.hip, DetectSink);
    extrema_place_kernel then orders slots and spill list alike.  A grid of dots puts 80 (short segments: one image) to
    160 (long segments: a batch of four) detections into one task: the raw list must still be the oracle's, byte for
    byte, with and without top-K."""
    imgs = np.stack([_dot_grid(248, 96, i) for i in range(4)])
    kw = dict(dog_threshold=0.0004, edge_threshold=60.0)
    for extra in ({}, dict(truncate_method=_abi.TRUNC_TOPK, feature_count_threshold=700)):
        g = gpu_ctx_factory(**kw, **extra)
        o = OracleSession(threads=8, keep_levels=False, **kw, **extra)
        for batch in (imgs[:1], imgs):
            _compare_all(g, o, batch, f"dense detections, batch of {len(batch)} {extra}", stages=False)
            if extra:   # (with top-K the oracle's list is the selected one; the scan's detections are the same as without)
                continue
            raw = o.rawlist(0)
            lvl0 = raw[raw["level_index"] < 3]   # octave 0: strips of 124 columns, segments of 12 (one or two images) or 24 rows
            rows = 12 if len(batch) <= 2 else 24
            task = (lvl0["col"] // 124) * 1000 + lvl0["row"] // rows
            assert np.bincount(task).max() > 64, "the input does not overflow a task's slots: the test would prove nothing"
        o.close()
