#!/usr/bin/env python3
"""Phase shares of the Gaussian kernel from a diagnostic build (-DHESS_GAUSS_STAMPS=1).  Round-3 tool: the stamps and the
column-strip march they were written for left k_gauss.hip in round 4; apply profiles/r03_experiments/gauss_march.diff
(patch -p0 from the repo root, against the round-3 file) to get a tree this script works on.

    python -m hessgpu_amd.build --variant stamps_tiles -DHESS_GAUSS_STAMPS=1 -DHESS_GAUSS_TILES=1
    python -m hessgpu_amd.build --variant stamps_march -DHESS_GAUSS_STAMPS=1
    HESS_LIB=tools/_variants/stamps_tiles/libhessgpu.so python tools/gauss_stamps.py [--octaves 1] [--batch 8]

Every wavefront adds the shader cycles it spends in each phase to a device table; printed: share of the summed
wavefront time per phase and mean cycles per wavefront.  The build forces waits at phase boundaries that the real
kernel leaves to the hardware: read the shares, not the run time."""
import argparse
import ctypes as C
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

PHASES = ["index math + load issue / bookkeeping", "waiting for source loads", "LDS stage stores + barrier",
          "fused det-H / gradient stage", "horizontal pass + barrier(s)", "vertical pass + store issue",
          "stores draining (tile form only)"]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--octaves", type=int, default=1)
    ap.add_argument("--batch", type=int, default=8)
    args = ap.parse_args()
    lib = hessgpu_amd.load_library()
    lib.hess_debug_gauss_stamps.argtypes = [C.POINTER(C.c_ulonglong)]
    imgs = np.stack([fixtures.synthetic_blobs(1920, 1080, i) for i in range(min(args.batch, 2))] * ((args.batch + 1) // 2))[:args.batch]
    d = torch.from_numpy(imgs).cuda()
    c = hessgpu_amd.HessContext(0, truncate_method=_abi.TRUNC_TOPK, feature_count_threshold=4096, octave_num=args.octaves)
    c.reserve(1920, 1080, args.batch)
    c.run_device(d.data_ptr(), args.batch, 1080, 1920)
    buf = (C.c_ulonglong * 8)()
    assert lib.hess_debug_gauss_stamps(buf) == 0
    for _ in range(5):
        c.run_device(d.data_ptr(), args.batch, 1080, 1920)
    assert lib.hess_debug_gauss_stamps(buf) == 0
    v = [buf[i] for i in range(8)]
    total, waves = float(sum(v[:7])), max(1, v[7])
    print(f"library {hessgpu_amd.LIB_PATH}: {waves} wavefronts over 5 runs, batch {args.batch}, {args.octaves} octave(s)")
    for i, name in enumerate(PHASES):
        print(f"  {name:44s} {100.0 * v[i] / total:5.1f} %   {v[i] / waves:9.0f} cycles per wavefront")
    print(f"  sum {total / waves:.0f} cycles per wavefront")


if __name__ == "__main__":
    main()
