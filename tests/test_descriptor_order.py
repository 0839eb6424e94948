"""The two descriptor summation orders of include/hess_abi.h (hess_params.descriptor_order).

SEQUENTIAL is the reference's order (ComputeDescriptor_Kernel, ProgramCU.cu:1723-1774: one thread per cell adds the
samples of its box as it walks them).  INTERLEAVED, the product's default, keeps four partial sums per bin (scan
positions 0..3 modulo 4) and adds them as (p0 + p1) + (p2 + p3).  The oracle restates both, so the HIP path is compared
BITWISE in either order; the two orders are tied to each other by a tolerance written here:

    TOL = 1e-6 on unit-norm descriptors (measured: <= 3e-7; the north star asks for 1e-4).

PIXEL (round 5) visits every pixel of the footprint once and adds its contributions in 32-bit fixed point (integer sums
do not depend on the order of the additions; oracle/hess_oracle.c: compute_descriptor_pixel).  Bitwise between the HIP
path and the oracle; against the reference's sequential order within

    TOL_PIXEL = 1e-5 on unit-norm descriptors (measured: <= 6e-6, most of it the sequential float order's own rounding:
    a float64 evaluation of the reference's formula is 7.9e-6 from the sequential order and 3.6e-7 from the pixel order
    for the worst feature of 640-2.jpg)."""
import numpy as np
import pytest

import fixtures
from oracle_lib import OracleSession

TOL = 1e-6
TOL_PIXEL = 1e-5
VARIANTS = [dict(), dict(half_sift=1), dict(dynamic_indexing=1), dict(normalize=0), dict(max_orientation=1)]


@pytest.mark.parametrize("kw", VARIANTS, ids=lambda k: "-".join(f"{a}{b}" for a, b in k.items()) or "default")
def test_oracle_orders_agree_within_tolerance(kw):
    img = fixtures.load_rgb("640-2.jpg")
    outs = []
    for order in (0, 1):
        o = OracleSession(threads=8, keep_levels=False, descriptor_order=order, **kw)
        o.run(img[None])
        outs.append(o.fetch(0))
        o.close()
    (k0, d0), (k1, d1) = outs
    assert k0.tobytes() == k1.tobytes() and len(k0) > 500          # the order touches descriptors only
    scale = 1.0 if kw.get("normalize", 1) else float(np.abs(d1).max())
    assert float(np.abs(d0 - d1).max()) <= TOL * scale
    assert (d0.view(np.uint32) != d1.view(np.uint32)).any()         # ... and they ARE two different summations


@pytest.mark.gpu
@pytest.mark.parametrize("kw", VARIANTS, ids=lambda k: "-".join(f"{a}{b}" for a, b in k.items()) or "default")
def test_gpu_sequential_order_is_bitwise_the_oracles(kw):
    import hessgpu_amd
    imgs = np.stack([fixtures.load_rgb(n)[..., 1] for n in ("640-1.jpg", "640-2.jpg", "640-3.jpg")])
    o = OracleSession(threads=16, keep_levels=False, descriptor_order=1, **kw)
    want = o.run(imgs)
    for batch in (imgs, imgs[:1]):            # copier delivery (descriptor_kernel<false, true>) and the host mirror (<true, true>)
        g = hessgpu_amd.HessContext(0, descriptor_order=1, **kw)
        assert g.run(batch) == want[:len(batch)]
        for i in range(len(batch)):
            gk, gd = g.fetch(i)
            ok, od = o.fetch(i)
            assert gk.tobytes() == ok.tobytes() and np.array_equal(gd.view(np.uint32), od.view(np.uint32)), (kw, i)
        g.close()
    # the interleaved order on the same images: equal keypoints, descriptors within the tolerance
    g = hessgpu_amd.HessContext(0, descriptor_order=0, **kw)
    g.run(imgs)
    for i in range(len(imgs)):
        gk, gd = g.fetch(i)
        ok, od = o.fetch(i)
        scale = 1.0 if kw.get("normalize", 1) else float(np.abs(od).max())
        assert gk.tobytes() == ok.tobytes() and float(np.abs(gd - od).max()) <= TOL * scale
    g.close()
    o.close()


@pytest.mark.gpu
def test_siftgpu_dseq_option():
    import siftgpu_lib
    img = np.ascontiguousarray(fixtures.load_rgb("640-1.jpg")[..., 1])
    s = siftgpu_lib.SiftGPU(["-dseq"])
    assert s.params().descriptor_order == 1
    assert s.run(img, siftgpu_lib.GL_LUMINANCE, siftgpu_lib.GL_UNSIGNED_BYTE) == 1
    k, d = s.features()
    o = OracleSession(threads=8, keep_levels=False, descriptor_order=1)
    o.run(img[None])
    ok, od = o.fetch(0)
    assert k.tobytes() == ok.tobytes() and np.array_equal(d.view(np.uint32), od.view(np.uint32))
    s.close()


@pytest.mark.parametrize("kw", VARIANTS, ids=lambda k: "-".join(f"{a}{b}" for a, b in k.items()) or "default")
def test_oracle_pixel_order_agrees_with_the_reference_order_within_tolerance(kw):
    img = fixtures.load_rgb("640-2.jpg")
    outs = []
    for order in (2, 1):
        o = OracleSession(threads=8, keep_levels=False, descriptor_order=order, **kw)
        o.run(img[None])
        outs.append(o.fetch(0))
        o.close()
    (k2, d2), (k1, d1) = outs
    assert k2.tobytes() == k1.tobytes() and len(k2) > 500
    scale = 1.0 if kw.get("normalize", 1) else float(np.abs(d1).max())
    assert float(np.abs(d2.astype(np.float64) - d1).max()) <= TOL_PIXEL * scale
    assert (d2.view(np.uint32) != d1.view(np.uint32)).any()


def test_oracle_pixel_order_keeps_float_orders_for_float_pixels_and_keypoint_lists():
    """The pixel order's fixed-point bound assumes luminance in [0, 1]: float pixels and user keypoint lists are
    described in the interleaved order (the same rule in hess_schedule.hip)."""
    lum = np.ascontiguousarray(fixtures.load_rgb("640-1.jpg")[..., 1])
    f32 = lum.astype(np.float32) / np.float32(255.0)
    res = []
    for order in (2, 0):
        o = OracleSession(threads=8, keep_levels=False, descriptor_order=order)
        o.run(f32[None])
        res.append(o.fetch(0))
        o.close()
    assert res[0][0].tobytes() == res[1][0].tobytes() and np.array_equal(res[0][1].view(np.uint32), res[1][1].view(np.uint32))


@pytest.mark.gpu
@pytest.mark.parametrize("kw", VARIANTS, ids=lambda k: "-".join(f"{a}{b}" for a, b in k.items()) or "default")
def test_gpu_pixel_order_is_bitwise_the_oracles(kw):
    import hessgpu_amd
    imgs = np.stack([fixtures.load_rgb(n)[..., 1] for n in ("640-1.jpg", "640-2.jpg", "640-3.jpg")])
    o = OracleSession(threads=16, keep_levels=False, descriptor_order=2, **kw)
    want = o.run(imgs)
    o1 = OracleSession(threads=16, keep_levels=False, descriptor_order=1, **kw)
    o1.run(imgs)
    for batch in (imgs, imgs[:1]):            # copier delivery (descriptor_pixel_kernel<false>) and the host mirror (<true>)
        g = hessgpu_amd.HessContext(0, descriptor_order=2, **kw)
        assert g.run(batch) == want[:len(batch)]
        for i in range(len(batch)):
            gk, gd = g.fetch(i)
            ok, od = o.fetch(i)
            assert gk.tobytes() == ok.tobytes() and np.array_equal(gd.view(np.uint32), od.view(np.uint32)), (kw, i)
            sk, sd = o1.fetch(i)
            scale = 1.0 if kw.get("normalize", 1) else float(np.abs(sd).max())
            assert gk.tobytes() == sk.tobytes() and float(np.abs(gd.astype(np.float64) - sd).max()) <= TOL_PIXEL * scale
        g.close()
    o.close()
    o1.close()
