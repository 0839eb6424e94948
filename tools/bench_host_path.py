#!/usr/bin/env python3
"""PCIe-inclusive rate of the hot path on ONE context: host pixels -> host results through hess_run_host (pageable
input: staged through the context's pinned buffer; pinned input: read by the copy engine directly) next to the
device-resident figure (hess_run_device).  Never the headline `value` (bench.py keeps the inputs resident in HBM and
reports the pipelined host-to-host rate as value_host_to_host); DESIGN.md section 6 quotes these numbers."""
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

W, H, TOPK = 1920, 1080, 4096


def rate(fn, B, n=20):
    for _ in range(3):
        fn()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    dt = time.perf_counter() - t0
    return round(n * B * W * H / dt / 1e6, 1), round(dt / n * 1e3, 3)


def main():
    out = {}
    for B in (16, 8, 1):
        imgs = np.ascontiguousarray(np.stack([fixtures.synthetic_blobs(W, H, i) for i in range(B)]))
        c = hessgpu_amd.HessContext(0, truncate_method=_abi.TRUNC_TOPK, feature_count_threshold=TOPK)
        c.reserve(W, H, B)
        d = torch.from_numpy(imgs).to("cuda:0")
        pin = torch.from_numpy(imgs).pin_memory()

        def pinned():
            c.submit_host(ptr=pin.data_ptr(), batch=B, height=H, width=W)
            c.wait()

        res = {"device_resident": rate(lambda: c.run_device(d.data_ptr(), B, H, W), B),
               "host_pageable": rate(lambda: c.run(imgs), B), "host_pinned": rate(pinned, B)}
        out[f"batch_{B}"] = {k: {"Mpix_per_s": v[0], "ms_per_batch": v[1]} for k, v in res.items()}
        c.close()
    print(json.dumps({"metric": "Mpixels/s, one context, host pixels -> host results vs device-resident input", **out}))


if __name__ == "__main__":
    main()
