#!/usr/bin/env python3
"""Where the time of a SHORT timed region goes (the driver runs bench.py --steps 20 --warmup 5): completion time of every
step relative to the start of the region, for several pipeline depths."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch

import fixtures
import hessgpu_amd
from hessgpu_amd import _abi

W, H, B, K = 1920, 1080, 8, int(sys.argv[1]) if len(sys.argv) > 1 else 20
WARM = [int(x) for x in sys.argv[2].split(",")] if len(sys.argv) > 2 else [5]
DEPTHS = [int(x) for x in sys.argv[3].split(",")] if len(sys.argv) > 3 else [2, 3, 4, 6]
PRE_MS = float(sys.argv[4]) if len(sys.argv) > 4 else 0.0
PACE = float(sys.argv[5]) if len(sys.argv) > 5 else 0.0
last_sub = [None]
RAMP = int(sys.argv[6]) if len(sys.argv) > 6 else 0
done = [0]
imgs = np.stack([fixtures.synthetic_blobs(W, H, i) for i in range(B)])
d = torch.from_numpy(imgs).cuda()
for nctx in DEPTHS:
    ctxs = [hessgpu_amd.HessContext(0, truncate_method=_abi.TRUNC_TOPK, feature_count_threshold=4096) for _ in range(nctx)]
    for c in ctxs:
        c.reserve(W, H, B)

    def run(n, stamps=None, t0=0.0):
        infl = []
        done[0] = 0
        for i in range(n):
            depth = nctx if (RAMP <= 0 or stamps is None) else min(nctx, RAMP + done[0])   # RAMP: initial depth, one more per completion
            free = [x for x in ctxs if x not in infl]
            c = free[0] if len(infl) < depth else None
            if len(infl) >= depth:
                c = infl.pop(0)
                c.wait()
                done[0] += 1
                if stamps is not None:
                    stamps.append(time.perf_counter() - t0)
            if PACE > 0 and last_sub[0] is not None:   # never two submissions closer than PACE ms
                while (time.perf_counter() - last_sub[0]) * 1e3 < PACE:
                    pass
            c.submit_device(d.data_ptr(), B, H, W)
            last_sub[0] = time.perf_counter()
            if stamps is not None:
                subs.append(time.perf_counter() - t0)
            infl.append(c)
        while infl:
            infl.pop(0).wait()
            if stamps is not None:
                stamps.append(time.perf_counter() - t0)

    for warm in WARM:
        # as bench.py does it: warm-up steps, fence, ONE timed region (a fresh idle gap before every measurement)
        time.sleep(0.5)
        if PRE_MS > 0:   # device conditioning: keep the GPU streaming HBM for PRE_MS before the warm-up steps
            big = torch.empty(256 << 20, dtype=torch.uint8, device="cuda")
            t_pre = time.perf_counter()
            while time.perf_counter() - t_pre < PRE_MS * 1e-3:
                for _ in range(8):
                    big.add_(1)
                torch.cuda.synchronize()
            del big
        run(warm)
        torch.cuda.synchronize()
        stamps, subs = [], []
        t0 = time.perf_counter()
        run(K, stamps, t0)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        gaps = np.diff([0.0] + stamps) * 1e3
        print(f"ramp {RAMP}, pace {PACE} ms, contexts {nctx}, {warm} warm-up steps after 0.5 s idle (+ {PRE_MS:.0f} ms of streaming): {K} steps in {dt*1e3:.2f} ms = {dt*1e3/K:.3f} ms/step = {B*K*W*H/dt/1e6:.0f} Mpix/s")
        print("   submit times (ms):", " ".join(f"{s_*1e3:.2f}" for s_ in subs[:8]), "...")
        print("   completion gaps (ms):", " ".join(f"{g:.2f}" for g in gaps))
    for c in ctxs:
        c.close()
