"""A caller compiled against the REFERENCE's header works unchanged against this build's libsiftgpu.so.

Container-side test (skipped where /root/reference is absent, e.g. on the GPU box): tests/abi/abi_caller.cpp is
compiled twice, with -I /root/reference/src/SiftGPU (SiftGPU.h:59-379 + config.h) and with -I include/, both
binaries are linked to hessgpu_amd/libsiftgpu.so and run.  Their outputs must be identical line by line -- sizes,
member offsets, enum values, and the results of calls made through the vtable / constructor symbols the respective
header declares (factory object as TestWin/SimpleSIFT.cpp:88-202, stack object as HessGPU/hessgpucmd.cpp:27)."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF_INC = "/root/reference/src/SiftGPU"
LIBDIR = os.path.join(ROOT, "hessgpu_amd")

pytestmark = pytest.mark.skipif(not os.path.exists(os.path.join(REF_INC, "SiftGPU.h")),
                                reason="reference header not present (only in the build container)")


def _build_and_run(tmp_path, inc, name):
    exe = str(tmp_path / name)
    subprocess.run(["g++", "-std=c++11", "-O0", "-I", inc, os.path.join(ROOT, "tests", "abi", "abi_caller.cpp"), "-o", exe,
                    "-L", LIBDIR, "-lsiftgpu", "-lhessgpu", f"-Wl,-rpath,{LIBDIR}"], check=True)
    r = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr[-2000:]
    return r.stdout.splitlines()


def test_caller_built_with_the_reference_header_runs_against_this_library(tmp_path):
    import hessgpu_amd.build as hb

    hb.build_all()
    ref = _build_and_run(tmp_path, REF_INC, "caller_ref")
    own = _build_and_run(tmp_path, os.path.join(ROOT, "include"), "caller_own")
    assert ref == own, "\n".join(f"{a!r} != {b!r}" for a, b in zip(ref, own) if a != b)
    d = dict(l.rsplit(" ", 1) for l in ref if " " in l)
    assert d["sizeof SiftParam"] == "56" and d["sizeof SiftGPU"] == "168" and d["sizeof SiftKeypoint"] == "24"
    assert d["offsetof _timing"] == "120" and d["sizeof timing"] == "48" and d["SIFT_KEYPOINT_ITEMS"] == "6"
    assert "factory ok" in ref and "GetImageCount 3" in ref and "stack GetImageCount 1" in ref
    assert "after ParseParam: _dog_level_num 4 _dog_threshold 0.0100" in ref   # -d 4 -t 0.01 landed in the SiftParam members
    assert "RunSIFT(missing file) 0" in ref and "IsFullSupported consistent 1" in ref
    assert "stack RunSIFT(pixels) returned" in ref and ref[-1] == "done"
    import torch

    assert ("CreateContextGL none" in ref) == (not torch.cuda.is_available())   # no GPU: loud 0, not a fallback
