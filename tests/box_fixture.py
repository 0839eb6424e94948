"""Analysis of the one feature file the reference ships, doc/evaluation/box.siftgpu (673 features of box.pgm,
made by demos/evaluation-box.bat: `-w 3 -fo -1 -loweo`), shared by tests/test_reference_fixture.py and
tests/golden/analyze_box_fixture.py.

The file comes from the DoG build of the family, so it cannot pin the Hessian DETECTOR.  It does pin everything
after detection, because both builds run the same code from a keypoint (x, y, scale) onwards and define the
Gaussian levels the keypoints are described at identically:

  DoG build     levels -1..4, sigma_l = 2.016 * 2^(l/3); keypoint list `level` (0..2) is described from the
                gradient plane of Gaussian level `level` (PyramidCU.cpp:1825-1846, 511-517 non-Hessian branches)
  Hessian build levels  0..4, sigma_l = 1.6   * 2^(l/3); detection levels 1..3 (SiftGPU.cpp:466-563)

2.016 = 1.6 * 2^(1/3): the DoG build's level l IS this build's level l+1, with the same inter-level blur
sqrt(sigma_l^2 - sigma_(l-1)^2) and the same level (sigma 3.2) decimated into the next octave.  The pyramids
coincide once the first octave is the up-sampled image (`-fo -1`: UpsampleKernel, ProgramCU.cu:233-310, then
G(sqrt(1.6^2 - 1.0^2))), which the C ABI and the oracle accept (the Hessian build refuses -fo < 0 only at its
option parser, SiftGPU.cpp:1166-1167).  So: build that pyramid, hand the file's keypoints to the user-keypoint
entry point (GenerateFeatureListTex, PyramidCU.cpp:555-718) and compare what comes out with the file.

Which level a file keypoint was described at is not in the file.  GenerateFeatureListTex bins a scale to the
level within half a step; a DETECTED keypoint is described at its detection level, whose sigma is further from
the refined scale (scale = sigma_level * step^ds).  In this file the level lies within (-2, +1) steps of the
scale's continuous level position (measured: 100 % of the interior keypoints), so each keypoint is tried at the
(at most) three levels floor(t), floor(t)+1, floor(t)+2 through the hess_debug_key_levels hook and the level
whose descriptor agrees is taken.  The choice is unambiguous: the right level agrees to <= 1 count of 512 in all
128 values, the runner-up is >= 12 counts off in every case.

What is left out, by rule and not by threshold: keypoints whose descriptor footprint (bounding circle of radius
2.5*sqrt(2)*3*scale around the centre) leaves the image.  What a sample outside the image contributes depends on
the backend that wrote the file (this path, the CUDA flavour, clamps the sample CENTRES to [1.5, W-1.5],
ProgramCU.cu:1723-1731; which backend and version wrote the file is not recorded), so those 92 keypoints are
reported, not asserted: 40 % of them still agree to <= 1 count, the ones whose rotated footprint stays inside.

The remaining <= 1 count is the file's own rounding: x, y to 0.01 px, scale and orientation to 0.001, and the
orientations of the multi-orientation path it was written with are 8-bit (2*pi/255 = 0.0246 rad).

Second, stronger use of the file (`reproduce_from_pixels`): the ORACLE can also run as the build the file was written
by -- `detector = 2` in hess_params: differences of Gaussians instead of det-Hessian planes, the extremum test without
its two sign conditions, the two-strongest-peaks orientation rule with 16-bit angles (everything the reference puts
under `#ifndef GPU_HESSIAN`: ProgramCU.cu:598-637,680-699,853-854,1493-1548; SiftGPU.cpp:466-556), and the level sigma
as it was before the "bug fix 9/12/2007" recorded at SiftGPU.cpp:1424 (the file's scales are sigma0 * 2^(level/6) *
step^ds; measured ratio to today's formula 2^(level/6) to four digits at every level).  Everything else is the SAME
oracle code the Hessian mode runs: pyramid, the 3x3x3 scan with its per-triple branch re-selection, edge test, the
pivoted 3x3 sub-pixel solve and its acceptance test, list order, keypoint packing and unpacking, orientation histogram,
multi-orientation expansion, descriptor, normalisation.  From the PIXELS of box.pgm it then finds 671 features at 539
locations (file: 673 at 541) and 664 of the file's 673 features are matched one to one within 0.024 px, 0.4 % in scale
and 0.001 rad for 654 of them (0.032 at most); every one of the 581 matched features whose footprint lies inside the
image has its descriptor within 1 count of 512 in all 128 values; 8 of the 9 unmatched features have their
orientation window cut by the image border.  This is the detector's reference-made evidence: the product is not run in
this mode (hess_create refuses `detector != 0`), it equals the oracle bit for bit in Hessian mode, and the two modes
differ only in the few lines listed above.
"""
import os

import numpy as np

from hessgpu_amd import _abi

_DATA = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "data")
W, H = 324, 223
PARAMS = dict(first_octave=-1, orient_window_factor=3.0, lowe_origin=1, max_orientation=1)
HALF_QUANTUM = np.pi / 255.0 + 0.0005   # half an 8-bit orientation step + the file's 3-decimal rounding
ONE_BIN = 2.0 * np.pi / 36.0            # one bin of the 36-bin orientation histogram


def load():
    from PIL import Image

    img = np.ascontiguousarray(np.asarray(Image.open(os.path.join(_DATA, "box.pgm"))))
    toks = open(os.path.join(_DATA, "box.siftgpu")).read().split()
    n, d = int(toks[0]), int(toks[1])
    vals = np.array(toks[2:], dtype=np.float64).reshape(n, 4 + d)   # y x scale orientation + 128 counts
    return img, vals


def keys_of(vals, dxy=0.0, flip=False):
    keys = np.zeros(len(vals), dtype=_abi.KEYPOINT_DTYPE)
    keys["y"], keys["x"], keys["s"] = vals[:, 0] + dxy, vals[:, 1] + dxy, vals[:, 2]
    keys["o"] = (2 * np.pi - vals[:, 3]) if flip else vals[:, 3]
    return keys


def level_position(scale):
    """Continuous level index t of a scale: sigma(li) = 0.5 * 1.6 * 2^((li+1)/3) with the first octave up-sampled."""
    return 3.0 * np.log2(scale / 0.8) - 1.0


def angle_diff(a, b):
    return np.abs(((a - b + np.pi) % (2 * np.pi)) - np.pi)


def analyse(session, img, vals):
    """Runs the file's keypoints through `session` (oracle or product, created with PARAMS) at each admissible
    level.  Returns a dict of per-keypoint arrays:
      err        worst of the 128 |floor(512 d + 0.5) - file| at the chosen level (counts of 512)
      err_next   the same at the best OTHER distinct level (how unambiguous the choice is)
      level, dlevel   chosen level index and t - level
      dangle     |computed strongest orientation - nearest file orientation of the same location|
      interior   descriptor footprint inside the image;  ointerior  orientation window inside the image
      desc       descriptors at the chosen level with the file's orientation (for bit comparisons between backends)
    """
    n = len(vals)
    session.run(img[None])
    nlev = len(session.geometry()) * 3
    t = level_position(vals[:, 2])
    keys = keys_of(vals)
    errs, angs, descs, levels = [], [], [], []
    for off in (0, 1, 2):
        lv = np.clip(np.floor(t).astype(np.int64) + off, 0, nlev - 1).astype(np.int32)
        session.debug_key_levels(lv)                            # (the hook covers one run: set again below)
        assert session.run_keypoints(keys, True) == n           # descriptors from the file's orientation
        d = session.fetch(0)[1]
        errs.append(np.abs(np.floor(512.0 * d + 0.5) - vals[:, 4:]).max(axis=1))   # SaveSIFT, SiftPyramid.cpp:504-566
        descs.append(d)
        session.debug_key_levels(lv)
        assert session.run_keypoints(keys, False) == n          # ComputeOrientation, strongest only (ProgramCU.cu:1398-1420)
        angs.append(session.fetch(0)[0]["o"].astype(np.float64))
        levels.append(lv)
    session.debug_key_levels(None)
    E, A, L = np.stack(errs, 1), np.stack(angs, 1), np.stack(levels, 1)
    pick = E.argmin(1)
    rows = np.arange(n)
    other = np.where(L != L[rows, pick][:, None], E, np.inf)
    ang = A[rows, pick]
    dangle = np.empty(n)
    for i in range(n):   # a location may be listed with several orientations: the strongest must be one of them
        same = (vals[:, 0] == vals[i, 0]) & (vals[:, 1] == vals[i, 1]) & (vals[:, 2] == vals[i, 2])
        dangle[i] = angle_diff(ang[i], vals[same, 3]).min()
    x, y, s = vals[:, 1], vals[:, 0], vals[:, 2]
    edge = np.minimum.reduce([x, W - x, y, H - y])
    return {
        "err": E[rows, pick], "err_next": other.min(1), "level": L[rows, pick], "dlevel": t - L[rows, pick],
        "dangle": dangle, "interior": edge - (2.5 * np.sqrt(2.0) * 3.0 * s + 1.0) >= 0,
        "ointerior": edge - (1.5 * 3.0 * s + 1.0) >= 0, "desc": np.stack(descs, 1)[rows, pick],
    }


DOG_PARAMS = dict(first_octave=-1, orient_window_factor=3.0, lowe_origin=1, detector=2)


def reproduce_from_pixels(session, img, vals):
    """Detect + describe box.pgm with `session` (oracle created with DOG_PARAMS) and match the result one to one with
    the file's features.  Returns a dict: n_features, n_locations, pairs (file index, result index), unmatched (file
    indices), and for the pairs: pos (px), angle (rad), scale_ratio, err (worst descriptor count difference),
    interior (descriptor footprint inside the image); for the unmatched: owin_margin (distance of the orientation
    window from the image border, negative = cut)."""
    n = session.run(img[None])[0]
    k, d = session.fetch(0)
    used = np.zeros(n, bool)
    pairs, unmatched = [], []
    for i in range(len(vals)):
        dist = np.hypot(k["x"] - vals[i, 1], k["y"] - vals[i, 0])
        da = angle_diff(k["o"].astype(np.float64), vals[i, 3])
        cand = np.flatnonzero((dist < 0.05) & (da < 0.05) & ~used)
        if len(cand):
            j = cand[np.argmin(dist[cand] + da[cand])]
            used[j] = True
            pairs.append((i, j))
        else:
            unmatched.append(i)
    pi, pj = np.array([p[0] for p in pairs]), np.array([p[1] for p in pairs])
    x, y, s = vals[:, 1], vals[:, 0], vals[:, 2]
    edge = np.minimum.reduce([x, W - x, y, H - y])
    return {
        "n_features": n, "n_locations": len(session.rawlist(0)), "pairs": pairs, "unmatched": unmatched,
        "pos": np.hypot(k["x"][pj] - x[pi], k["y"][pj] - y[pi]),
        "angle": angle_diff(k["o"][pj].astype(np.float64), vals[pi, 3]),
        "scale_ratio": k["s"][pj] / s[pi],
        "err": np.abs(np.floor(512.0 * d[pj] + 0.5) - vals[pi, 4:]).max(axis=1),
        "interior": (edge - (2.5 * np.sqrt(2.0) * 3.0 * s + 1.0) >= 0)[pi],
        "owin_margin": (edge - (1.5 * 3.0 * s + 1.0))[np.array(unmatched, dtype=int)],
    }
