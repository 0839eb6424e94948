#!/usr/bin/env python3
"""A/B of the descriptor kernel's accumulation, as it was run: the sequential order (bit-identical to
the oracle's reference mode) against the lane-private form, then a variant build (-DHESS_DESC_PRIVATE_BINS, run as
HESS_LIB=tools/_variants/desc_private/libhessgpu.so) -- since adopted as hess_params.descriptor_order = INTERLEAVED, the
default.  Prints the largest difference of any descriptor value between the library under test (default order) and the
oracle's SEQUENTIAL order on three reference images and on bench image 0 (the north star's tolerance is 1e-4)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, fixtures, hessgpu_amd
from hessgpu_amd import _abi
from oracle_lib import OracleSession
worst = 0.0
cases = [fixtures.load_rgb(n)[..., 1].copy() for n in ("640-1.jpg", "640-2.jpg", "640-3.jpg")] + [fixtures.synthetic_blobs(1920, 1080, 0)]
for img in cases:
    kw = dict(truncate_method=_abi.TRUNC_TOPK, feature_count_threshold=4096)
    g = hessgpu_amd.HessContext(0, **kw); o = OracleSession(threads=16, keep_levels=False, descriptor_order=1, **kw)
    ng, no = g.run(img[None]), o.run(img[None])
    gk, gd = g.fetch(0); ok, od = o.fetch(0)
    assert ng == no and gk.tobytes() == ok.tobytes(), "keypoints differ"
    d = float(np.abs(gd - od).max()); nbits = int((gd.view(np.uint32) != od.view(np.uint32)).sum())
    worst = max(worst, d)
    print(f"{img.shape}: {ng[0]} features, max |desc - oracle| = {d:.3e}, values that differ in any bit: {nbits} of {gd.size}")
    g.close(); o.close()
print("worst", worst, "tolerance 1e-4:", "ok" if worst <= 1e-4 else "EXCEEDED")
