#!/usr/bin/env python3
"""One-time costs of a context's first batches: fresh contexts, reserve, then synchronous runs timed one by one."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import fixtures, hessgpu_amd
from hessgpu_amd import _abi
W, H, B = 1920, 1080, 8
imgs = np.stack([fixtures.synthetic_blobs(W, H, i) for i in range(B)])
d = torch.from_numpy(imgs).cuda()
torch.cuda.synchronize()
for k in range(4):
    t = time.perf_counter()
    c = hessgpu_amd.HessContext(0, truncate_method=_abi.TRUNC_TOPK, feature_count_threshold=4096)
    t1 = time.perf_counter()
    c.reserve(W, H, B)
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    runs = []
    for r in range(5):
        a = time.perf_counter()
        c.run_device(d.data_ptr(), B, H, W)
        runs.append((time.perf_counter() - a) * 1e3)
    print(f"context {k}: create {1e3*(t1-t):.1f} ms, reserve {1e3*(t2-t1):.1f} ms, runs (ms): " + " ".join(f"{x:.2f}" for x in runs))
    globals()[f"keep{k}"] = c
