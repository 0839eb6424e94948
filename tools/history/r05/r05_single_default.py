#!/usr/bin/env python3
"""One image per run, default parameters (no top-K: the capacity-based delivery rule decides): ms per image by size."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import fixtures, hessgpu_amd
from hessgpu_amd import _abi
for (W, H, kw) in ((640, 480, {}), (1920, 1080, {}), (1920, 1080, dict(truncate_method=_abi.TRUNC_TOPK, feature_count_threshold=4096)),
                   (2560, 1920, {}), (4096, 4096, dict(tex_max_dim=4096))):
    img = fixtures.synthetic_blobs(W, H, 0)
    d = torch.from_numpy(img[None]).to("cuda:0")
    c = hessgpu_amd.HessContext(0, **kw)
    c.reserve(W, H, 1)
    for _ in range(3):
        c.run_device(d.data_ptr(), 1, H, W)
    n = c.count(0)
    t0 = time.perf_counter()
    for _ in range(30):
        c.run_device(d.data_ptr(), 1, H, W)
    dt = (time.perf_counter() - t0) / 30
    print(f"{W}x{H} {kw and 'topk' or 'default'}: {n} features, {dt * 1e3:.3f} ms per image", flush=True)
    c.close()
