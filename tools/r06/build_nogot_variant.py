#!/usr/bin/env python3
"""EXPERIMENT library (not the product): tools/_variants/x_nogot/libhessgpu.so -- from a context's third batch on the
Gaussian launches no longer write the gradient/theta planes (the bench re-runs the same images, so the planes of the
first batches stay valid and the results stay right): what a pipeline whose planes were free would run at -- the upper
bound of what computing them only under the features' footprints can gain."""
import os
import shutil
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from hessgpu_amd import build

tmp = tempfile.mkdtemp(prefix="hess_nogot_")
dst = os.path.join(tmp, "hessgpu_amd", "csrc")
os.makedirs(os.path.dirname(dst))
shutil.copytree(build.CSRC, dst, ignore=shutil.ignore_patterns("_obj"))
shutil.copytree(os.path.join(ROOT, "include"), os.path.join(tmp, "include"))
p = os.path.join(dst, "hess_schedule.hip")
s = open(p).read()
old = "    j.norm_src = s.norm[l - 1];\n"
new = "    if (c->x_batches > 2) j.got_src = nullptr;\n    j.norm_src = s.norm[l - 1];\n"
assert old in s
s = s.replace(old, new, 1)
old = "  c->stage_events = (c->p.verbose & 2) != 0;\n"
assert old in s
s = s.replace(old, old + "  c->x_batches++;\n", 1)
open(p, "w").write(s)
p = os.path.join(dst, "hess_ctx.h")
s = open(p).read()
old = "  int regrown = 0;"
assert old in s
s = s.replace(old, "  int x_batches = 0;\n  int regrown = 0;", 1)
open(p, "w").write(s)
build.CSRC = dst
print(build.build_variant("x_nogot", [], verbose=True))
shutil.rmtree(tmp)
