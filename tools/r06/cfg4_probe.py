#!/usr/bin/env python3
"""BASELINE.json configs[4] (one 4096x4096 image, -maxd 4096 -topk 65536 -half) on one context: per-kernel hipEvent
times per image, the descriptor launches' feature footprints, one and three contexts.  One JSON line.
  --runs N      profiled runs (default 5)        --quick   no pipelined leg (counter passes)
  --delivery X  HESS_DELIVERY for the context (dma: four launches over quarters of the list, as a submitted image gets)"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
if "--delivery" in sys.argv:
    os.environ["HESS_DELIVERY"] = sys.argv[sys.argv.index("--delivery") + 1]
import numpy as np
import torch

import fixtures
import hessgpu_amd
from hessgpu_amd import _abi

S = 4096


def main():
    runs = int(sys.argv[sys.argv.index("--runs") + 1]) if "--runs" in sys.argv else 5
    img = fixtures.synthetic_blobs(S, S, 0)
    d = torch.from_numpy(img[None]).to("cuda:0")
    nctx = 1 if "--quick" in sys.argv else (int(sys.argv[sys.argv.index("--contexts") + 1]) if "--contexts" in sys.argv else 3)
    ctxs = [hessgpu_amd.HessContext(0, tex_max_dim=4096, half_sift=1, truncate_method=_abi.TRUNC_TOPK,
                                    feature_count_threshold=65536) for _ in range(nctx)]
    for c in ctxs:
        c.reserve(S, S, 1)
        c.run_device(d.data_ptr(), 1, S, S)
    c = ctxs[0]
    keys = c.fetch(0)[0]
    out = {"features": int(c.count(0))}
    if nctx >= 2:
        t0 = time.perf_counter()
        for _ in range(10):
            c.run_device(d.data_ptr(), 1, S, S)
        out["ms_one_context"] = round((time.perf_counter() - t0) / 10 * 1e3, 3)
        steps, inflight = 60, []
        t0 = time.perf_counter()
        for i in range(steps):
            x = ctxs[i % nctx]
            if len(inflight) == nctx:
                inflight.pop(0).wait()
            x.submit_device(d.data_ptr(), 1, S, S)
            inflight.append(x)
        while inflight:
            inflight.pop(0).wait()
        dt = (time.perf_counter() - t0) / steps
        out["ms_three_contexts"] = round(dt * 1e3, 3)
        out["Mpix_per_s_three_contexts"] = round(S * S / dt / 1e6, 1)
    c.profile_enable(True)
    c.profile_reset()
    for _ in range(runs):
        c.run_device(d.data_ptr(), 1, S, S)
    prof = c.profile()
    out["kernel_ms_per_image"] = {k: round(v["ms"] / runs, 4) for k, v in prof.items() if v["launches"]}
    out["descriptor_launches_per_image"] = prof["descriptor"]["launches"] // runs
    # footprint (bounding box of the rotated 5x5-cell window, pixels) per feature, as the kernel forms it
    s_oct = keys["s"].astype(np.float64) / (2.0 ** (keys["level"] // 3))
    spt = 3.0 * s_oct
    o = keys["o"].astype(np.float64)
    side = 2.0 * 2.5 * spt * (np.abs(np.cos(o)) + np.abs(np.sin(o))) + 1.0
    steps_f = np.ceil(side * side / 64.0)
    out["footprint"] = {"mean_side_px": round(float(side.mean()), 1), "max_side_px": round(float(side.max()), 1),
                        "mean_steps_of_64px": round(float(steps_f.mean()), 1), "max_steps": int(steps_f.max()),
                        "side_percentiles_10_50_90_99": [round(float(x), 1) for x in np.percentile(side, [10, 50, 90, 99])],
                        "features_by_octave": np.bincount(keys["level"] // 3).tolist()}
    for x in ctxs:
        x.close()
    print(json.dumps(out))


if __name__ == "__main__":
    main()
