"""The `hess` batch tool and the `speed` harness (apps/, reference: src/HessGPU/hessgpucmd.cpp,
src/TestWin/speed.cpp) end to end on the GPU."""
import os
import subprocess

import numpy as np
import pytest

import fixtures
from oracle_lib import OracleSession

pytestmark = pytest.mark.gpu
BIN = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "hessgpu_amd", "bin")


def _write_pgm(path, lum):
    with open(path, "wb") as f:
        f.write(b"P5\n%d %d\n255\n" % (lum.shape[1], lum.shape[0]))
        f.write(lum.tobytes())


def test_hess_cli_list_mode_writes_sift_and_timings(tmp_path):
    names = ["640-4.jpg", "640-5.jpg"]
    lums = [fixtures.load_rgb(n)[..., 1].copy() for n in names]
    for n, l in zip(names, lums):
        _write_pgm(tmp_path / (n[:-4] + ".pgm"), l)
    (tmp_path / "list.txt").write_text("640-4.pgm\n640-5.pgm\n")
    r = subprocess.run([os.path.join(BIN, "hess"), "-il", str(tmp_path / "list.txt"), "-time", "-topk", "300"],
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    o = OracleSession(threads=8, keep_levels=False, truncate_method=3, feature_count_threshold=300)
    for n, l in zip(names, lums):
        base = tmp_path / (n[:-4] + ".pgm")
        tok = open(str(base) + ".sift").read().split()
        o.run(l[None])
        k, d = o.fetch(0)
        assert int(tok[0]) == len(k) and int(tok[1]) == 128
        first = [int(v) for v in tok[2 + 7: 2 + 7 + 128]]
        assert first == [int(np.floor(0.5 + 512.0 * v)) for v in d[0]]
        t = [float(v) for v in open(str(base) + ".timings").read().split(",")]
        assert len(t) == 11 and t[-1] > 0 and t[2] > 0  # total, pyramid


def test_hess_cli_usage_without_images():
    r = subprocess.run([os.path.join(BIN, "hess")], capture_output=True, text=True, timeout=60)
    assert r.returncode != 0 and "-il" in r.stdout


def test_speed_harness_reports_stable_counts(tmp_path):
    lum = fixtures.load_rgb("640-1.jpg")[..., 1].copy()
    _write_pgm(tmp_path / "a.pgm", lum)
    r = subprocess.run([os.path.join(BIN, "speed"), "-i", str(tmp_path / "a.pgm"), "-n", "5"], capture_output=True,
                       text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    # two passes as in the reference (speed.cpp:107-152): '+' per repetition without stage timers, '#' with them
    marks = [l for l in r.stdout.splitlines() if l and set(l) <= set("+#e")]
    assert marks == ["+++++", "#####"], r.stdout
    vals = {l.split("]:")[0].strip("["): float(l.split("]:")[1].replace("ms per image", "").replace("ms", "").replace("hz", ""))
            for l in r.stdout.splitlines() if l.startswith("[")}
    o = OracleSession(threads=8, keep_levels=False)
    o.run(lum[None])
    assert vals["Feature Count"] == o.count(0) and vals["Average Speed"] > 0
    for k in ("Build Pyramid", "Detection", "Feature List", "Orientation", "Descriptor"):
        assert vals[k] > 0, (k, r.stdout)                  # stage timers were on in the second pass
    assert vals["With stage timers"] >= 0.8 * vals["Average Time"]


def test_multigpu_driver_gathers_on_device_0(tmp_path):
    """apps/multigpu.cpp: one host thread per device through the C ABI, exact-size RCCL send / recv of the feature lists
    to device 0, the gathered copy compared with every device's own host results.  On this box: every visible device
    (one), i.e. the count table, the ncclGroup and the check without a peer."""
    import torch
    names = ["640-1.jpg", "640-2.jpg", "640-3.jpg"]
    lums = [fixtures.load_rgb(n)[..., 1].copy() for n in names]
    args = [os.path.join(BIN, "multigpu"), "-n", "3", "-batch", "3", "-topk", "500"]
    for n, l in zip(names, lums):
        _write_pgm(tmp_path / (n[:-4] + ".pgm"), l)
        args += ["-i", str(tmp_path / (n[:-4] + ".pgm"))]
    r = subprocess.run(args, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    ndev = torch.cuda.device_count()
    o = OracleSession(threads=8, keep_levels=False, truncate_method=3, feature_count_threshold=500)
    want = o.run(np.stack(lums))
    rows = [l for l in r.stdout.splitlines() if l.startswith("#")]
    assert len(rows) == ndev
    for d, l in enumerate(rows):          # "#d: n n n features": device d holds images 3d .. 3d+2 of the cycled list
        assert [int(v) for v in l.split(":")[1].split()[:-1]] == [want[(3 * d + b) % 3] for b in range(3)]
    last = r.stdout.splitlines()[-1]
    assert last.startswith("GATHER OK") and int(last.split()[2]) == ndev * sum(want)


def test_multithread_driver_one_instance_per_thread_per_device(tmp_path):
    """apps/multithread.cpp, the reference's multi-GPU pattern (TestWin/MultiThreadSIFT.cpp:83-156,231-244): one
    SiftGPU instance per host thread per device, initialised under a mutex, RunSIFT() repeated without a lock.
    Runs on every visible device (one here, all of them on a multi-GPU node), two instances per device."""
    names = ["640-1.jpg", "640-3.jpg"]
    lums = [fixtures.load_rgb(n)[..., 1].copy() for n in names]
    args = [os.path.join(BIN, "multithread"), "-n", "20", "-per-device", "2"]
    for n, l in zip(names, lums):
        _write_pgm(tmp_path / (n[:-4] + ".pgm"), l)
        args += ["-i", str(tmp_path / (n[:-4] + ".pgm"))]
    r = subprocess.run(args, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    lines = [l for l in r.stdout.splitlines() if l.startswith("#")]
    import torch

    assert len(lines) == 2 * torch.cuda.device_count() and r.stdout.splitlines()[-1].startswith("OK")
    o = OracleSession(threads=8, keep_levels=False)
    want = {}
    for n, l in zip(names, lums):
        o.run(l[None])
        want[n[:-4] + ".pgm"] = o.count(0)
    for l in lines:                      # "#t: device d, path: N features, H Hz"
        path, rest = l.split(", ", 1)[1].split(": ")
        assert int(rest.split()[0]) == want[os.path.basename(path)] and float(rest.split(", ")[1].split()[0]) > 0
