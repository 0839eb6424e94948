#!/usr/bin/env python3
"""PCIe-inclusive rate of the hot path: host pixels -> host results through hess_run_host (one H2D copy
of the u8 images per batch on top of what bench.py times).  Never the headline `value` (bench.py keeps
the inputs resident in HBM); DESIGN.md section 6 quotes this number next to it."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np

import fixtures
import hessgpu_amd
from hessgpu_amd import _abi

W, H, TOPK, B = 1920, 1080, 4096, 16


def main():
    imgs = np.stack([fixtures.synthetic_blobs(W, H, i) for i in range(4)])
    imgs = np.ascontiguousarray(np.concatenate([imgs] * 4)[:B])
    c = hessgpu_amd.HessContext(0, truncate_method=_abi.TRUNC_TOPK, feature_count_threshold=TOPK)
    c.reserve(W, H, B)
    for _ in range(3):
        c.run(imgs)
    n, t0 = 0, time.perf_counter()
    while n < 20:
        c.run(imgs)
        n += 1
    dt = time.perf_counter() - t0
    print(json.dumps({"metric": "Mpixels/s host pixels -> host results (hess_run_host, pageable numpy input, one context)",
                      "value": round(n * B * W * H / dt / 1e6, 1), "ms_per_batch_of_16": round(dt / n * 1e3, 3)}))
    c.close()


if __name__ == "__main__":
    main()
