#!/usr/bin/env python3
"""EXPERIMENT library from a patched copy of the kernel sources (the tree is not touched):
   tools/r06/build_patched_variant.py NAME FILE OLD NEW [FILE OLD NEW ...] [-- compiler flags]
-> tools/_variants/NAME/libhessgpu.so (select with HESS_LIB=...; tools/r06/ab_lib.sh, desc_ab.sh, kstat_lib.sh)."""
import os
import shutil
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from hessgpu_amd import build

args = sys.argv[1:]
flags = []
if "--" in args:
    flags = args[args.index("--") + 1:]
    args = args[:args.index("--")]
name, edits = args[0], args[1:]
assert len(edits) % 3 == 0
tmp = tempfile.mkdtemp(prefix="hess_patch_")
dst = os.path.join(tmp, "hessgpu_amd", "csrc")
os.makedirs(os.path.dirname(dst))
shutil.copytree(build.CSRC, dst, ignore=shutil.ignore_patterns("_obj"))
shutil.copytree(os.path.join(ROOT, "include"), os.path.join(tmp, "include"))
for f, old, new in zip(edits[0::3], edits[1::3], edits[2::3]):
    p = os.path.join(dst, f)
    s = open(p).read()
    assert s.count(old) == 1, (f, old, s.count(old))
    open(p, "w").write(s.replace(old, new))
build.CSRC = dst
print(build.build_variant(name, flags))
shutil.rmtree(tmp)
