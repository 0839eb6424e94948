#!/usr/bin/env python3
"""SiftGPU::RunSIFT + GetFeatureVector on one large image (configs[4]'s) through the class, pageable pixels: ms per image."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import fixtures, siftgpu_lib
for (S, args) in ((4096, ["-maxd", "4096", "-topk", "65536", "-half"]), (2560, ["-maxd", "4096"])):
    img = fixtures.synthetic_blobs(S, S if S == 4096 else 1920, 0)
    s = siftgpu_lib.SiftGPU(args)
    for _ in range(3):
        assert s.run(img, siftgpu_lib.GL_LUMINANCE, siftgpu_lib.GL_UNSIGNED_BYTE) == 1
    k, d = s.features()
    t0 = time.perf_counter()
    for _ in range(10):
        s.run(img, siftgpu_lib.GL_LUMINANCE, siftgpu_lib.GL_UNSIGNED_BYTE)
        k, d = s.features()
    dt = (time.perf_counter() - t0) / 10
    print(f"{img.shape[1]}x{img.shape[0]} {' '.join(args)}: {len(k)} features, {dt * 1e3:.3f} ms per image (RunSIFT + GetFeatureVector)", flush=True)
    s.close()
