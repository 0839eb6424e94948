"""bench.py prints ONE JSON line with the contract's keys (driver contract + roofline + cpu_baseline)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_json_line():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "3", "--warmup", "1",
                        "--batch", "2"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, r.stdout
    d = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "roofline_secondary", "cpu_baseline",
              "value_host_to_host", "latency_ms_single_image", "parity_checked", "kernel_ms_per_step"):
        assert k in d, k
    assert d["unit"] == "Mpix/s" and d["n_gpus"] == 1 and d["steps"] == 3 and d["warmup"] == 1
    assert d["higher_is_better"] is True and d["scaling"] == "weak" and d["vs_baseline"] is None
    assert d["data"] == "synthetic" and d["dtype"] == "f32" and "workload" in d["config"]
    assert d["value"] > 0 and abs(d["value"] - 2 * 3 * 1920 * 1080 / (d["ms_per_step"] * 3 * 1e-3) / 1e6) / d["value"] < 0.01
    rf = d["roofline"]
    assert rf["bound"] == "hbm" and rf["unit"] == "GB/s" and rf["peak"] == 8000.0
    assert abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-3 and 0 < rf["frac"] < 1
    kernels = {rf["kernel"].split()[0], d["roofline_secondary"]["kernel"].split()[0]}
    assert kernels == {"gauss_kernel", "descriptor_kernel<true>"} and rf["ms_per_step"] >= d["roofline_secondary"]["ms_per_step"]
    dk = rf if rf["kernel"].startswith("descriptor") else d["roofline_secondary"]
    assert dk["without_host_mirror"]["kernel"] == "descriptor_kernel<false>" and dk["without_host_mirror"]["avg_launch_us"] > 0
    assert d["parity_checked"] is True                       # image 0 of the timed run == the oracle, bit for bit
    assert 0 < d["value_host_to_host"] and 0 < d["latency_ms_single_image"] < 100
    assert d["config"]["distinct_images_per_gpu"] == 2 and "configs[1]" in d["config"]["workload"]
    cb = d["cpu_baseline"]
    assert cb["kind"] == "port" and cb["unit"] == "Mpix/s" and cb["cores"] >= 1 and cb["value"] > 0 and cb["sample"]
