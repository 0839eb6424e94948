#!/usr/bin/env python3
"""EXPERIMENT libraries: tools/_variants/x_grid<N>/libhessgpu.so -- the descriptor launch's grid = N / 8 wavefronts per
feature of the last batch (product: 10 / 8).  tools/r06/desc_ab.sh TAG x_grid4 x_grid6 x_grid8 cur x_grid14."""
import os
import shutil
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from hessgpu_amd import build

for n in [int(a) for a in sys.argv[1:]] or [4, 6, 8, 14]:
    tmp = tempfile.mkdtemp(prefix="hess_grid_")
    dst = os.path.join(tmp, "hessgpu_amd", "csrc")
    os.makedirs(os.path.dirname(dst))
    shutil.copytree(build.CSRC, dst, ignore=shutil.ignore_patterns("_obj"))
    shutil.copytree(os.path.join(ROOT, "include"), os.path.join(tmp, "include"))
    p = os.path.join(dst, "k_feature.hip")
    s = open(p).read()
    old = "(long long)seen_features * 5 / 4 / den"
    assert s.count(old) == 1
    open(p, "w").write(s.replace(old, f"(long long)seen_features * {n} / 8 / den"))
    keep = build.CSRC
    build.CSRC = dst
    print(build.build_variant(f"x_grid{n}", []))
    build.CSRC = keep
    shutil.rmtree(tmp)
