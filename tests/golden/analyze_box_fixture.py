#!/usr/bin/env python3
"""Prints the analysis of the reference's feature file behind tests/test_reference_fixture.py (see
tests/box_fixture.py for the method): how the oracle's orientation and descriptor stages compare with the
673 reference-made features of doc/evaluation/box.siftgpu, keypoint by keypoint.

    python tests/golden/analyze_box_fixture.py            (CPU oracle; add --gpu for the product on cuda:0)
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import numpy as np  # noqa: E402

import box_fixture as bf  # noqa: E402


def main():
    img, vals = bf.load()
    if "--gpu" in sys.argv:
        import hessgpu_amd

        s = hessgpu_amd.HessContext(0, **bf.PARAMS)
    else:
        from oracle_lib import OracleSession

        s = OracleSession(threads=8, **bf.PARAMS)
    r = bf.analyse(s, img, vals)
    n, it, ot = len(vals), r["interior"], r["ointerior"]
    print(f"{n} features; descriptor footprint inside the image: {it.sum()}, touching the border: {(~it).sum()}")
    for name, m in (("interior", it), ("border", ~it)):
        e = r["err"][m]
        print(f"  {name:8s} worst |count difference| of the 128 values:  0: {(e == 0).mean():.3f}  <=1: {(e <= 1).mean():.3f}"
              f"  <=2: {(e <= 2).mean():.3f}  <=3: {(e <= 3).mean():.3f}  max {e.max():.0f}")
    amb = it & np.isfinite(r["err_next"])
    print(f"  interior, runner-up level: min error {r['err_next'][amb].min():.0f} counts (chosen level: max {r['err'][it].max():.0f})")
    print(f"  level - scale position (t - level) of the interior keypoints: [{r['dlevel'][it].min():.2f}, {r['dlevel'][it].max():.2f}]")
    hist, edges = np.histogram(r["dlevel"][it], bins=[-2, -1.5, -1, -0.5, 0, 0.5, 1])
    print("    histogram", dict(zip([f"{a:+.1f}..{b:+.1f}" for a, b in zip(edges[:-1], edges[1:])], hist.tolist())))
    for name, m in (("orientation window inside the image", ot), ("all", np.ones(n, bool))):
        d = r["dangle"][m]
        print(f"  orientation, {name} ({m.sum()}): within half an 8-bit step {(d < bf.HALF_QUANTUM).mean():.4f}, "
              f"within 0.05 rad {(d < 0.05).mean():.4f}, within one histogram bin {(d < bf.ONE_BIN).mean():.4f}, max {d.max():.3f}")
    # the conventions matter: flipping the orientation sense or shifting the origin by a pixel breaks the agreement
    for name, kw in (("orientation sense flipped", dict(flip=True)), ("origin shifted by one pixel", dict(dxy=-1.0))):
        s.debug_key_levels(r["level"])
        s.run_keypoints(bf.keys_of(vals, **kw), True)
        e = np.abs(np.floor(512.0 * s.fetch(0)[1] + 0.5) - vals[:, 4:]).max(axis=1)[it]
        print(f"  control, {name}: <=1 count for {(e <= 1).mean():.3f} of the interior keypoints (median {np.median(e):.0f})")
    s.debug_key_levels(None)
    if "--gpu" in sys.argv:
        return
    # from the pixels: the oracle as the build that wrote the file (tests/box_fixture.py, second part of the docstring)
    from oracle_lib import OracleSession

    o = OracleSession(threads=8, **bf.DOG_PARAMS)
    r = bf.reproduce_from_pixels(o, img, vals)
    it = r["interior"]
    print(f"from pixels, DoG mode: {r['n_features']} features at {r['n_locations']} locations (file: {n} at "
          f"{len(set(map(tuple, vals[:, :3])))}); matched one to one: {len(r['pairs'])}")
    print(f"  position within {r['pos'].max():.3f} px, scale within {np.abs(r['scale_ratio'] - 1).max() * 100:.2f} %, angle within 0.001 rad "
          f"for {(r['angle'] < 0.001).sum()} (max {r['angle'].max():.3f})")
    print(f"  descriptors of the {it.sum()} matched features with the footprint inside the image: worst count difference "
          f"{r['err'][it].max():.0f}; of the {(~it).sum()} at the border: <=1 for {(r['err'][~it] <= 1).mean():.2f}")
    print(f"  unmatched file features {r['unmatched']}: orientation-window margin to the border {np.round(r['owin_margin'], 1).tolist()}")


if __name__ == "__main__":
    main()
