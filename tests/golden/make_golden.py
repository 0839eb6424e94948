#!/usr/bin/env python3
"""Generates tests/golden/oracle_golden.json: digests of the CPU oracle's outputs on the
reference's data/ images (copied under tests/golden/data) and on the synthetic generator.

The reference ships no golden vectors for the Hessian path and cannot be run here,
so these vectors pin the ORACLE against unintended change; they are not reference outputs.
Run from the repo root:  python tests/golden/make_golden.py
"""
import hashlib
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import numpy as np  # noqa: E402

import fixtures  # noqa: E402
from oracle_lib import OracleSession  # noqa: E402

CASES = [
    ("640-1.jpg", {}),
    ("640-2.jpg", {}),
    ("640-3.jpg", {}),
    ("640-4.jpg", {}),
    ("640-5.jpg", {}),
    ("blobs.png", {}),
    ("sunflowers.png", {}),
    ("640-1.jpg", {"half_sift": 1}),
    ("640-1.jpg", {"truncate_method": 3, "feature_count_threshold": 100}),
    ("640-1.jpg", {"max_orientation": 1}),
    ("synthetic:640x360:7", {"truncate_method": 3, "feature_count_threshold": 512}),
]


def digest(name, kw):
    if name.startswith("synthetic:"):
        _, size, idx = name.split(":")
        w, h = (int(v) for v in size.split("x"))
        img = fixtures.synthetic_blobs(w, h, int(idx))
    else:
        img = fixtures.load_rgb(name)
    o = OracleSession(threads=4, keep_levels=False, descriptor_order=1, **kw)  # the reference's (sequential) summation order
    n = o.run(img[None])[0]
    k, d = o.fetch(0)
    raw = o.rawlist(0)
    return {
        "image": name, "params": kw, "input_sha256": hashlib.sha256(img.tobytes()).hexdigest(),
        "features": int(n), "locations": int(len(raw)),
        "types": [int((k["type"] == t).sum()) for t in range(3)],
        "levels": np.bincount(k["level"], minlength=1).tolist(),
        "keys_sha256": hashlib.sha256(k.tobytes()).hexdigest(),
        "desc_sha256": hashlib.sha256(d.tobytes()).hexdigest(),
        "first_keys": [[float(k[f][i]) for f in ("x", "y", "s", "o", "response")] for i in range(min(4, n))],
        "first_desc_head": [float(v) for v in d[0][:8]] if n else [],
    }


if __name__ == "__main__":
    out = [digest(n, kw) for n, kw in CASES]
    with open(os.path.join(ROOT, "tests", "golden", "oracle_golden.json"), "w") as f:
        json.dump(out, f, indent=1)
    print("wrote", len(out), "cases")
