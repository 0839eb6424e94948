"""The C-ABI library loads on a machine without a GPU and exports every symbol that
include/hess_abi.h declares; host-side logic that needs no device behaves as documented."""
import ctypes as C
import os
import re

import pytest

import hessgpu_amd
from hessgpu_amd import _abi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_functions(header):
    text = open(header).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(hess_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    lib = hessgpu_amd.load_library()
    names = _declared_functions(os.path.join(ROOT, "include", "hess_abi.h"))
    assert len(names) >= 19 and "hess_run_device" in names and "hess_fetch" in names
    missing = [n for n in names if not hasattr(lib, n)]
    assert not missing, f"libhessgpu.so lacks {missing}"


def test_struct_layouts_match_the_reference_records():
    assert C.sizeof(_abi.HessKeypoint) == 24      # SiftGPU::SiftKeypoint, SiftGPU.h:108-116
    assert C.sizeof(_abi.HessRawKey) == 32
    assert C.sizeof(_abi.HessParams) == 4 * 24 + 4 * 8
    assert _abi.HessKeypoint.level.offset == 20 and _abi.HessKeypoint.type.offset == 22


def test_default_params_are_the_reference_defaults():
    p = hessgpu_amd.default_params()
    assert p.abi_version == _abi.HESS_ABI_VERSION
    assert p.dog_level_num == 3 and abs(p.sigma0 - 1.6) < 1e-7 and abs(p.sigman - 0.5) < 1e-7
    assert abs(p.dog_threshold - 0.02 / 3) < 1e-9 and p.edge_threshold == 10.0
    assert p.filter_width_factor == 4.0 and p.orient_window_factor == 2.0 and p.desc_window_factor == 3.0
    assert p.subpixel == 1 and p.max_orientation == 2 and p.tex_max_dim == 3200
    assert p.feature_count_threshold == -1 and p.compute_descriptors == 1 and p.normalize == 1


def test_no_gpu_means_loud_failure_not_fallback():
    import torch

    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(hessgpu_amd.HessError):
        hessgpu_amd.HessContext(0)


def test_product_does_not_reference_the_oracle():
    """The product path must not import, link or call anything under oracle/."""
    pkg = os.path.join(ROOT, "hessgpu_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                text = open(os.path.join(dirpath, f), errors="ignore").read()
                assert "hess_oracle" not in text and "oracle_lib" not in text and "hess_cpu_" not in text.replace(
                    "`hess_cpu_`", ""), f"{f} refers to the oracle"
    import subprocess

    so = os.path.join(pkg, "libhessgpu.so")
    out = subprocess.run(["ldd", so], capture_output=True, text=True).stdout
    assert "hess_oracle" not in out
