#!/usr/bin/env python3
"""Host-to-host time of ONE 1080p image on one context (hess_run_device from resident pixels, hess_run_host from pageable
pixels), median of 400 calls: tools/r06/lat_ab.sh runs it per library variant (HESS_LIB)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch, fixtures, hessgpu_amd
from hessgpu_amd import _abi
img = fixtures.synthetic_blobs(1920, 1080, 0)[None]
c = hessgpu_amd.HessContext(0, truncate_method=_abi.TRUNC_TOPK, feature_count_threshold=4096)
c.reserve(1920, 1080, 1)
d = torch.from_numpy(img).to("cuda:0")
def med(f, n=400):
    for _ in range(20): f()
    t = []
    for _ in range(n):
        t0 = time.perf_counter(); f(); t.append(time.perf_counter() - t0)
    t.sort(); return round(t[len(t) // 2] * 1e3, 4), round(t[len(t) // 10] * 1e3, 4)
print(os.environ.get("HESS_LIB", "cur").split("/")[-2] if os.environ.get("HESS_LIB") else "cur",
      "device pixels (median, p10) ms", med(lambda: c.run_device(d.data_ptr(), 1, 1080, 1920)),
      "host pixels", med(lambda: c.run(img)), "features", c.count(0))
c.close()
