"""The two descriptor summation orders of include/hess_abi.h (hess_params.descriptor_order).

SEQUENTIAL is the reference's order (ComputeDescriptor_Kernel, ProgramCU.cu:1723-1774: one thread per cell adds the
samples of its box as it walks them).  INTERLEAVED, the product's default, keeps four partial sums per bin (scan
positions 0..3 modulo 4) and adds them as (p0 + p1) + (p2 + p3).  The oracle restates both, so the HIP path is compared
BITWISE in either order; the two orders are tied to each other by a tolerance written here:

    TOL = 1e-6 on unit-norm descriptors (measured: <= 3e-7; the north star asks for 1e-4)."""
import numpy as np
import pytest

import fixtures
from oracle_lib import OracleSession

TOL = 1e-6
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
    # the default order on the same images: equal keypoints, descriptors within the tolerance (and bitwise the oracle's
    # default order: every other parity test)
    g = hessgpu_amd.HessContext(0, **kw)
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
