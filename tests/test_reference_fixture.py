"""The one feature file the reference ships -- doc/evaluation/box.siftgpu, 673 features of box.pgm in SaveSIFT's
text format ("y x scale orientation" + 128 values floor(512 d + 0.5)), made by demos/evaluation-box.bat with
`-w 3 -fo -1 -loweo` -- against this build's orientation and descriptor stages.  Method and the reasoning behind
each rule: tests/box_fixture.py; printed analysis: tests/golden/analyze_box_fixture.py.

Asserted, for the oracle on the CPU and for the HIP path on the GPU (which must also equal the oracle bit for bit):
  * descriptors (ComputeDescriptor + NormalizeDescriptor + SaveSIFT quantisation) computed from the file's
    (x, y, scale, orientation): EVERY keypoint whose footprint lies inside the image (581 of 673) agrees with the
    file to <= 1 count of 512 in all 128 values, at a level that is unambiguous (runner-up >= 10 counts off);
  * orientations (ComputeOrientation, strongest peak, ProgramCU.cu:1398-1420) computed from (x, y, scale) alone:
    >= 99 % of the keypoints whose window lies inside the image are within half an 8-bit orientation step of an
    orientation the file lists for that location, all of them within one histogram bin;
  * the coordinate and angle conventions matter: flipping the angle sense or moving the origin by one pixel
    leaves no keypoint in agreement.
The remaining 92 keypoints touch the image border (rule in tests/box_fixture.py) and are not asserted.
And from the PIXELS: the oracle run as the build that wrote the file (DoG planes, extremum test without the two sign
conditions, two-peak orientation rule -- the reference's `#ifndef GPU_HESSIAN` lines -- on otherwise the same code)
finds the file's features again: detection, sub-pixel refinement, orientation and descriptor end to end.
What stays UNPINNED: the Hessian-specific lines themselves (det-H formula, the sign conditions, the blob/saddle type)
and top-K: no reference output of the Hessian path exists."""
import numpy as np
import pytest

import box_fixture as bf
from oracle_lib import OracleSession


def _assert_pinned(r, vals):
    it, ot = r["interior"], r["ointerior"]
    assert it.sum() == 581 and ot.sum() == 650
    assert r["err"][it].max() <= 1.0, np.flatnonzero(it & (r["err"] > 1))          # counts of 512, all 128 values
    amb = it & np.isfinite(r["err_next"])
    assert amb.sum() > 500 and r["err_next"][amb].min() >= 10.0                     # the level choice is not a fit
    assert -2.0 < r["dlevel"][it].min() and r["dlevel"][it].max() < 1.0             # admissible levels only
    d = r["dangle"][ot]
    assert (d < bf.HALF_QUANTUM).mean() >= 0.99 and d.max() < bf.ONE_BIN, ((d < bf.HALF_QUANTUM).mean(), d.max())
    assert (r["err"][~it] <= 1).mean() > 0.3                                        # border keypoints: reported only


@pytest.mark.parametrize("order", [0, 1], ids=["interleaved", "sequential"])  # hess_params.descriptor_order: both pinned by the file
def test_oracle_orientation_and_descriptors_match_the_reference_file(order):
    img, vals = bf.load()
    assert img.shape == (bf.H, bf.W) and vals.shape == (673, 132)
    ref = vals[:, 4:] / 512.0
    assert np.all(np.abs(np.linalg.norm(ref, axis=1) - 1.0) < 0.02)      # unit descriptors, quantised
    o = OracleSession(threads=8, descriptor_order=order, **bf.PARAMS)
    r = bf.analyse(o, img, vals)
    _assert_pinned(r, vals)
    # controls: the same keypoints with the angle sense flipped / the origin moved by a pixel agree nowhere
    for kw in (dict(flip=True), dict(dxy=-1.0)):
        o.debug_key_levels(r["level"])
        o.run_keypoints(bf.keys_of(vals, **kw), True)
        e = np.abs(np.floor(512.0 * o.fetch(0)[1] + 0.5) - vals[:, 4:]).max(axis=1)[r["interior"]]
        assert (e <= 1).mean() < 0.02 and np.median(e) > 30
    o.close()


def test_oracle_as_the_dog_build_reproduces_the_file_from_pixels():
    img, vals = bf.load()
    o = OracleSession(threads=8, **bf.DOG_PARAMS)
    r = bf.reproduce_from_pixels(o, img, vals)
    o.close()
    assert abs(r["n_features"] - 673) <= 3 and abs(r["n_locations"] - 541) <= 3, (r["n_features"], r["n_locations"])
    assert len(r["pairs"]) >= 0.985 * len(vals)                       # 664 of 673 matched one to one
    assert r["pos"].max() < 0.03 and np.abs(r["scale_ratio"] - 1.0).max() < 0.005
    assert (r["angle"] < 0.0015).mean() > 0.98 and r["angle"].max() < 0.05
    it = r["interior"]
    assert it.sum() >= 575 and r["err"][it].max() <= 1.0              # every interior descriptor within 1 count of 512
    assert (r["owin_margin"] < 0).sum() >= len(r["unmatched"]) - 1    # what is not matched sits at the image border


def test_border_keypoints_of_the_file_follow_the_packed_glsl_rule():
    """The 92 keypoints whose descriptor footprint touches the image border: with the CUDA path's clamp (sample centres in
    [1.5, dim-1.5], ProgramCU.cu:1723-1731 -- the product's rule) 40 % of them agree with the file, with the clamp of the
    PACKED GLSL shaders (box clamped to [2, dim-3] and widened to whole 2x2 texels, ProgramGLSL.cpp:2579-2581; an
    analysis switch of the oracle, never of the product) 85 % do, and the orientations of the cut windows follow.  So the
    file was written by the packed GLSL backend: its border keypoints cannot pin the CUDA path's border rule, and they are
    no evidence against it either; the interior ones are the same under both rules."""
    img, vals = bf.load()
    res = {}
    for border in (0, 1):
        o = OracleSession(threads=8, border=border, **bf.PARAMS)
        res[border] = bf.analyse(o, img, vals)
        o.close()
    it = res[0]["interior"]
    assert it.sum() == 581 and (~it).sum() == 92
    for border in (0, 1):
        assert (res[border]["err"][it] <= 1).all()                       # interior keypoints: the rule does not matter
    cuda, packed = (res[0]["err"][~it] <= 1).sum(), (res[1]["err"][~it] <= 1).sum()
    assert cuda <= 45 and packed >= 75, (cuda, packed)
    cut = ~res[0]["ointerior"]
    assert (res[1]["dangle"][cut] < bf.HALF_QUANTUM).mean() > (res[0]["dangle"][cut] < bf.HALF_QUANTUM).mean() + 0.2


def test_dog_mode_is_refused_by_the_product():
    import ctypes as C

    import hessgpu_amd

    hessgpu_amd.load_library()
    p = hessgpu_amd.default_params()
    p.reserved[0] = 2            # the oracle's detector word: the product wants every reserved word zero
    assert not hessgpu_amd.functions()["create"](0, C.byref(p))


def test_default_level_binning_without_the_hook():
    """Without the level hook GenerateFeatureListTex's half-step rule picks the level (PyramidCU.cpp:597-601):
    where that is the file's level the result is the file's again."""
    img, vals = bf.load()
    o = OracleSession(threads=8, **bf.PARAMS)
    r = bf.analyse(o, img, vals)
    o.run_keypoints(bf.keys_of(vals), True)
    e = np.abs(np.floor(512.0 * o.fetch(0)[1] + 0.5) - vals[:, 4:]).max(axis=1)
    same = r["interior"] & (np.abs(r["dlevel"]) < 0.5)     # the half-step rule lands on the chosen level
    assert same.sum() > 250 and e[same].max() <= 1.0
    o.close()


@pytest.mark.gpu
def test_gpu_orientation_and_descriptors_match_the_reference_file(gpu_ctx_factory):
    img, vals = bf.load()
    g = gpu_ctx_factory(**bf.PARAMS)
    o = OracleSession(threads=8, **bf.PARAMS)
    rg, ro = bf.analyse(g, img, vals), bf.analyse(o, img, vals)
    _assert_pinned(rg, vals)
    assert np.array_equal(rg["level"], ro["level"])
    assert np.array_equal(rg["desc"].view(np.uint32), ro["desc"].view(np.uint32))   # HIP == oracle, bit for bit
    assert np.array_equal(rg["dangle"], ro["dangle"])
    # the up-sampled first octave itself (UpsampleKernel + first blur) equals the oracle's
    g.run(img[None]); o.run(img[None])
    for octave in range(len(o.geometry())):
        for level in (0, 3):
            assert np.array_equal(g.level(0, octave, level, 0).view(np.uint32), o.level(0, octave, level, 0).view(np.uint32))
    o.close()
