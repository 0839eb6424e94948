#!/usr/bin/env python3
"""Throughput of the descriptor matcher (hess_matcher_*) on the GPU, with the CPU
oracle timed beside it on a smaller problem.  Prints one JSON line per size."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np

from hessgpu_amd.matcher import Matcher


def main():
    rng = np.random.RandomState(0)
    sizes = (1024, 4096, 8192)
    if "--only" in sys.argv:
        sizes = (int(sys.argv[sys.argv.index("--only") + 1]),)
    for n in sizes:
        d1 = (rng.rand(n, 128) * 45).astype(np.uint8)
        d2 = (rng.rand(n, 128) * 45).astype(np.uint8)
        m = Matcher(0, max_sift=n)
        m.set_descriptors(0, d1)
        m.set_descriptors(1, d2)
        m.match(max_match=n)
        ts = []
        for _ in range(10):
            m.match(max_match=n)
            ts.append(m.last_ms())
        ms = float(np.median(ts))
        out = {"n1": n, "n2": n, "device_ms": round(ms, 4), "GMAC_per_s": round(n * n * 128 / ms / 1e6, 1),
               "path": ("unguided: v_mfma_i32_32x32x32_i8 tiles, row/column reductions folded in registers, no score "
                        "matrix in memory") if n * n > (3 << 20) else "unguided, small: one-pass v_dot4 tiles + score matrix"}
        if n == 1024:
            from oracle_lib import oracle_match

            t0 = time.perf_counter()
            oracle_match(d1, d2, max_match=n)
            out["cpu_oracle_ms"] = round((time.perf_counter() - t0) * 1e3, 1)
        print(json.dumps(out), flush=True)
        m.close()


if __name__ == "__main__":
    main()
