#!/usr/bin/env python3
"""BASELINE.json configs[4]: one 4096x4096 synthetic image, -maxd 4096 -topk 65536 -half (64-d descriptors),
device-resident input, results to host memory; prints one JSON line (not the headline metric)."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch

import fixtures
import hessgpu_amd
from hessgpu_amd import _abi

W = H = 4096


def main():
    img = fixtures.synthetic_blobs(W, H, 0)   # the generator scales its blob count with the area
    d = torch.from_numpy(img[None]).to("cuda:0")
    ctxs = [hessgpu_amd.HessContext(0, tex_max_dim=4096, half_sift=1, truncate_method=_abi.TRUNC_TOPK,
                                    feature_count_threshold=65536) for _ in range(3)]
    for c in ctxs:
        c.reserve(W, H, 1)
        c.run_device(d.data_ptr(), 1, H, W)
    n = ctxs[0].count(0)
    # one context, synchronous (latency of one image)
    t0 = time.perf_counter()
    for _ in range(10):
        ctxs[0].run_device(d.data_ptr(), 1, H, W)
    lat = (time.perf_counter() - t0) / 10
    # three contexts pipelined
    steps, inflight = 30, []
    t0 = time.perf_counter()
    for i in range(steps):
        c = ctxs[i % 3]
        if len(inflight) == 3:
            inflight.pop(0).wait()
        c.submit_device(d.data_ptr(), 1, H, W)
        inflight.append(c)
    while inflight:
        inflight.pop(0).wait()
    dt = (time.perf_counter() - t0) / steps
    print(json.dumps({"workload": "4096x4096 synthetic, -maxd 4096 -topk 65536 -half", "features": n,
                      "ms_per_image_one_context": round(lat * 1e3, 3), "Mpix_per_s_one_context": round(W * H / lat / 1e6, 1),
                      "ms_per_image_three_contexts": round(dt * 1e3, 3), "Mpix_per_s_three_contexts": round(W * H / dt / 1e6, 1)}))


if __name__ == "__main__":
    main()
