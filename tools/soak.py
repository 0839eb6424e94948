#!/usr/bin/env python3
"""Determinism soak: the same batch through three pipelined contexts N times; every run must give the same
keypoints and descriptors bit for bit (atomics only ever touch positional masks and counters)."""
import hashlib
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch

import fixtures
import hessgpu_amd
from hessgpu_amd import _abi


def digest(c, B):
    h = hashlib.sha256()
    for b in range(B):
        k, d = c.fetch(b)
        h.update(k.tobytes())
        h.update(d.tobytes())
    return h.hexdigest()


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 150
    host = len(sys.argv) > 2 and sys.argv[2] == "host"   # pinned host pixels (hess_submit_host: upload beside the queues, kernels enqueued by the copier thread)
    B, W, H = 8, 1920, 1080
    imgs = np.stack([fixtures.synthetic_blobs(W, H, i) for i in range(4)] * 2)
    d = torch.from_numpy(imgs).to("cuda:0")
    pinned = torch.from_numpy(imgs).pin_memory()
    ctxs = [hessgpu_amd.HessContext(0, truncate_method=_abi.TRUNC_TOPK, feature_count_threshold=4096) for _ in range(3)]
    ref = None
    inflight = []
    bad = 0
    for i in range(n):
        c = ctxs[i % 3]
        if len(inflight) == 3:
            f = inflight.pop(0)
            f.wait()
            dg = digest(f, B)
            ref = ref or dg
            bad += dg != ref
        if host:
            c.submit_host(ptr=pinned.data_ptr(), batch=B, height=H, width=W)
        else:
            c.submit_device(d.data_ptr(), B, H, W)
        inflight.append(c)
    while inflight:
        f = inflight.pop(0)
        f.wait()
        bad += digest(f, B) != ref
    print(f"soak: {n} pipelined batches of {B} images ({'pinned host' if host else 'device-resident'} pixels), {bad} differ from the first")
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
