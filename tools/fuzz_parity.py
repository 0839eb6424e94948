#!/usr/bin/env python3
"""Randomised parity sweep on the GPU: random image sizes, contents and parameters, every stage compared bit for bit
with the CPU oracle (the comparison of tests/test_gpu_parity.py).  Not part of the test suite (run time grows with
the number of cases):   python tools/fuzz_parity.py [cases=40] [seed=1]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np

import fixtures
import hessgpu_amd
from hessgpu_amd import _abi
from oracle_lib import OracleSession
from test_gpu_parity import _compare_all


def main():
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    rng = np.random.RandomState(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
    for k in range(cases):
        w, h = int(rng.randint(40, 1300)), int(rng.randint(40, 700))
        batch = int(rng.choice([1, 1, 2, 3]))
        kind = rng.choice(["noise", "blobs", "smooth"])
        if kind == "noise":
            img = (rng.rand(batch, h, w) * 255).astype(np.uint8)
        elif kind == "blobs":
            img = np.stack([fixtures.synthetic_blobs(w, h, int(rng.randint(1000))) for _ in range(batch)])
        else:
            yy, xx = np.mgrid[0:h, 0:w]
            img = np.stack([(127 + 120 * np.sin(xx / rng.uniform(3, 40)) * np.cos(yy / rng.uniform(3, 40))).astype(np.uint8)
                            for _ in range(batch)])
        kw = {}
        if rng.rand() < 0.5:
            kw.update(truncate_method=_abi.TRUNC_TOPK, feature_count_threshold=int(rng.choice([10, 300, 4096])))
        elif rng.rand() < 0.3:
            kw.update(truncate_method=int(rng.choice([_abi.TRUNC_HIGHEST_0, _abi.TRUNC_HIGHEST_1, _abi.TRUNC_LOWEST])),
                      feature_count_threshold=int(rng.choice([50, 500])))
        if rng.rand() < 0.3:
            kw["half_sift"] = 1
        if rng.rand() < 0.3:
            kw["max_orientation"] = int(rng.choice([1, 3, 4]))
        if rng.rand() < 0.3:
            kw["dog_level_num"] = int(rng.choice([1, 2, 4, 5]))
        if rng.rand() < 0.3:
            kw["dog_threshold"] = float(rng.choice([0.0005, 0.004, 0.03]))
        if rng.rand() < 0.2:
            kw["first_octave"] = 1
        if rng.rand() < 0.2:
            kw["subpixel"] = 0
        if rng.rand() < 0.4:
            kw["descriptor_order"] = int(rng.choice([0, 1]))    # interleaved / the reference's sequential order (default: pixel raster)
        if rng.rand() < 0.15:
            kw["dynamic_indexing"] = 1
        if rng.rand() < 0.15:
            kw["normalize"] = 0
        g = hessgpu_amd.HessContext(0, **kw)
        o = OracleSession(threads=16, **kw)
        try:
            n = _compare_all(g, o, img, f"case {k}: {w}x{h}x{batch} {kind} {kw}", stages=True)
        finally:
            g.close()
            o.close()
        print(f"case {k}: {w}x{h}x{batch} {kind} {kw} -> {n} features ok", flush=True)
    print(f"{cases} cases identical")


if __name__ == "__main__":
    main()
