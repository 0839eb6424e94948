#!/usr/bin/env python3
"""What a reference-style caller pays after RunSIFT: hess_fetch of one 1080p image's results (5.6 k keypoints, 2.9 MB of
descriptors) from the context's pinned result buffers into the caller's arrays, beside the run itself (median of 300)."""
import ctypes as C, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, fixtures, hessgpu_amd
from hessgpu_amd import _abi
img = fixtures.synthetic_blobs(1920, 1080, 0)[None]
c = hessgpu_amd.HessContext(0, truncate_method=_abi.TRUNC_TOPK, feature_count_threshold=4096)
c.reserve(1920, 1080, 1)
c.run(img)
n = c.count(0)
keys = np.zeros(n, dtype=_abi.KEYPOINT_DTYPE); desc = np.zeros((n, 128), dtype=np.float32)
kp, dp = keys.ctypes.data_as(C.c_void_p), desc.ctypes.data_as(C.c_void_p)
tr, tf = [], []
for i in range(320):
    t0 = time.perf_counter(); c.run(img); t1 = time.perf_counter(); c._f["fetch"](c._h, 0, kp, dp); t2 = time.perf_counter()
    if i >= 20: tr.append(t1 - t0); tf.append(t2 - t1)
tr.sort(); tf.sort()
print("features", n, "run ms", round(tr[len(tr)//2]*1e3, 4), "fetch ms", round(tf[len(tf)//2]*1e3, 4), "MB", round(n*(128*4+24)/1e6, 2),
      "GB/s", round(n*(128*4+24)/tf[len(tf)//2]/1e9, 1))
ok, od = c.fetch(0)
assert ok.tobytes() == keys.tobytes() and od.tobytes() == desc.tobytes()
c.close()
