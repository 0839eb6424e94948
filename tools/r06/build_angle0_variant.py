#!/usr/bin/env python3
"""EXPERIMENT library (not the product; its descriptors are WRONG): tools/_variants/x_angle0/libhessgpu.so -- the pixel
descriptor kernel takes every feature's orientation as 0, so the raster's bounding box IS the 5 x 5-cell window (no
lanes outside it).  Upper bound of what a raster over the rotated window's own row spans can gain over the bounding
box (mean box / window area over uniform angles: 1 + 2/pi = 1.64)."""
import os
import shutil
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from hessgpu_amd import build

tmp = tempfile.mkdtemp(prefix="hess_angle0_")
dst = os.path.join(tmp, "hessgpu_amd", "csrc")
os.makedirs(os.path.dirname(dst))
shutil.copytree(build.CSRC, dst, ignore=shutil.ignore_patterns("_obj"))
shutil.copytree(os.path.join(ROOT, "include"), os.path.join(tmp, "include"))
p = os.path.join(dst, "k_feature.hip")
s = open(p).read()
old = "    const float spt = fabsf(kz * dp.window_factor);\n    float s, c;\n    dm_sincosf(kw, &s, &c);\n    const float anglef"
assert s.count(old) == 1, s.count(old)
s = s.replace(old, "    const float spt = fabsf(kz * dp.window_factor);\n    float s, c;\n    dm_sincosf(kw * 0.0f, &s, &c);\n    const float anglef", 1)
open(p, "w").write(s)
build.CSRC = dst
print(build.build_variant("x_angle0", [], verbose=True))
shutil.rmtree(tmp)
