#!/usr/bin/env python3
"""One 1080p image through one context, repeated (for rocprofv3 --kernel-trace + tools/trace_timeline.py)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch, fixtures, hessgpu_amd
from hessgpu_amd import _abi
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
imgs = np.stack([fixtures.synthetic_blobs(1920, 1080, i) for i in range(B)])
c = hessgpu_amd.HessContext(0, truncate_method=_abi.TRUNC_TOPK, feature_count_threshold=4096)
c.reserve(1920, 1080, B)
d = torch.from_numpy(imgs).to("cuda:0")
for _ in range(30):
    c.run_device(d.data_ptr(), B, 1080, 1920)
c.close()
