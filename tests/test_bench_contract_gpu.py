"""bench.py prints ONE JSON line with the contract's keys (driver contract + roofline + cpu_baseline)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _fracs(node, path=""):
    """Every fraction-of-a-peak on the line, wherever it sits."""
    if isinstance(node, dict):
        for k, v in node.items():
            if k in ("frac", "frac_of_measured", "frac_of_mix") and v is not None:
                yield path + "/" + k, v
            else:
                yield from _fracs(v, path + "/" + k)
    elif isinstance(node, list):
        for i, v in enumerate(node):
            yield from _fracs(v, f"{path}[{i}]")


def test_bench_json_line():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "3", "--warmup", "1",
                        "--batch", "3"], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, r.stdout
    d = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "roofline_secondary", "cpu_baseline",
              "value_host_to_host", "value_device_resident", "latency_ms_single_image", "parity_checked",
              "kernel_ms_per_step", "value_siftgpu_api_1thread", "value_siftgpu_api_threads", "configs4", "real_images", "matcher"):
        assert k in d, k
    assert d["unit"] == "Mpix/s" and d["n_gpus"] == 1 and d["steps"] == 3 and d["warmup"] == 1
    assert d["higher_is_better"] is True and d["scaling"] == "weak" and d["vs_baseline"] is None
    assert d["data"] == "synthetic" and d["dtype"] == "f32" and "workload" in d["config"]
    assert d["value"] > 0 and abs(d["value"] - 3 * 3 * 1920 * 1080 / (d["ms_per_step"] * 3 * 1e-3) / 1e6) / d["value"] < 0.01
    assert d["value_device_resident"] == d["value"]
    assert d["value_steady_state"] > 0.8 * d["value"]   # 200 steps of the same pipeline (a 3-step region is mostly fill and drain)
    rf = d["roofline"]
    assert rf["bound"] == "hbm" and rf["unit"] == "GB/s" and rf["peak"] == 8000.0
    assert abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-3 and 0 < rf["frac"] < 1
    kernels = {rf["kernel"].split(" (")[0].split(" /")[0], d["roofline_secondary"]["kernel"].split(" (")[0].split(" /")[0]}
    # a batch of three images is delivered by the copier thread: the descriptor launch does not mirror to the host
    # (names as in a kernel trace: descriptor_pixel_kernel<host mirror>, the default order's kernel)
    assert kernels == {"gauss_kernel", "descriptor_pixel_kernel<false>"} and rf["ms_per_step"] >= d["roofline_secondary"]["ms_per_step"]
    dk = rf if rf["kernel"].startswith("descriptor") else d["roofline_secondary"]
    assert dk["with_host_mirror"]["kernel"] == "descriptor_pixel_kernel<true>" and dk["with_host_mirror"]["avg_launch_us"] > 0
    if "committed_profiles_stale" not in d:   # (the vector-issue entries come from a committed PMC pass: left off when that pass
        assert dk["valu"]["peak"] == 1228.8 and 0 < dk["valu"]["frac"] < 1   #  was of other kernel sources than this run's)
    gk = rf if rf["kernel"].startswith("gauss") else d["roofline_secondary"]
    o0 = gk["octave0_launches"]  # the Gaussian launches of octave 0, timed apart: the bandwidth-bound part of the stage
    assert o0["launches"] > 0 and 0 < o0["frac"] < 1 and o0["frac"] > gk["frac"] and 0.5 < o0["share_of_stage_bytes"] < 1
    # no number on the line may exceed the peak it is a fraction of (round 3 published a vector-issue frac of 15.94)
    fr = dict(_fracs(d))
    assert len(fr) >= 8, fr
    for where, v in fr.items():
        assert 0.0 < v <= 1.0, (where, v)
    # the dominant kernel of the COMMITTED kernel trace (a verbatim copy of profiles/kernel_stats_top.json, marked as
    # such: not a measurement of this run)
    rr = d["roofline_by_rocprof"]
    assert rr["kernel"].startswith(("descriptor", "gauss")) and rr["avg_launch_us"] > 0 and rr["source"].startswith("profiles/")
    assert rr["from_committed_profile"] is True
    assert ("stale" in rr) == ("committed_profiles_stale" in d and "kernel_stats_top.json" in d["committed_profiles_stale"]["files"])
    # the legs of round 6: the reference's own photographs and the matcher
    ri = d["real_images"]
    big = next(v for k, v in ri.items() if k.startswith("1600.jpg"))
    assert big["Mpix_per_s_three_contexts"] > 0 and big["features_per_image_mean"] > 1000 and "gauss" in big["kernel_ms_per_step"]
    mt = d["matcher"]
    assert mt["bound"] == "mfma_i8" and 0 < mt["frac"] < 1 and mt["parity_checked"] is True and mt["8192x8192"]["TMAC_per_s"] > 0
    assert d["parity_checked"] is True                       # image 0 of the timed run == the oracle, bit for bit
    par = d["parity"]   # ... in the SAME summation order bit for bit, and against the reference's order within the tolerance
    assert par["bitwise_vs_oracle_in_the_same_order"] is True and par["keypoints_bitwise_vs_oracle_in_the_reference_order"] is True
    assert par["descriptor_max_abs_diff_vs_reference_order"] <= par["tolerance_vs_reference_order"] <= 1e-5
    assert d["config"]["descriptor_order"] in ("interleaved", "sequential", "pixel")
    assert 0 < d["value_host_to_host"] and 0 < d["latency_ms_single_image"] < 100
    assert d["config"]["distinct_images_per_gpu"] == 3 and "configs[1]" in d["config"]["workload"]
    cb = d["cpu_baseline"]
    assert cb["kind"] == "port" and cb["unit"] == "Mpix/s" and cb["cores"] >= 1 and cb["value"] > 0 and cb["sample"]
    # the plugin surface (libsiftgpu.so: RunSIFT(w,h,data) + GetFeatureVector), one and several instances
    assert d["value_siftgpu_api_1thread"] > 0 and d["value_siftgpu_api_threads"] > 0
    c4 = d["configs4"]
    assert "configs[4]" in c4["workload"] and c4["features"] >= 65536
    assert c4["Mpix_per_s_one_context"] > 0 and c4["Mpix_per_s_three_contexts"] > 0
    r4 = c4["roofline_descriptor"]
    assert r4["bound"] == "hbm" and r4["peak"] == 8000.0 and 0 < r4["frac"] < 1 and abs(r4["features_per_launch"] * r4["launches_per_image"] - c4["features"]) <= r4["launches_per_image"]


def test_driver_command_is_close_to_steady_state():
    """The driver's own command (--steps 20 --warmup 5): the contract's fences put the pipeline's fill and drain inside the
    20 timed steps, but no context may run its FIRST batch there (round 4's driver run lost 21 % = 3.4 ms that way: seven
    contexts, five warm-up steps).  What the region costs beyond 20 steady-state steps is 1.5 ms on most boxes of the pool
    and 2.7 ms on some (`value` 20.7 against 19.0 Gpix/s at the same steady state, tools/r06/driver_vs_steady.sh): bounded
    by 4 ms here.  One repeat is allowed: a stall of several milliseconds inside the region was seen once in about sixty
    runs (tools/r06/driver_outliers.sh) and is not what this test is about."""
    over = []
    for attempt in range(2):
        r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "20", "--warmup", "5",
                            "--no-configs4", "--no-api-leg", "--no-host-leg", "--no-cpu-baseline", "--no-real-images", "--no-matcher"],
                           capture_output=True, text=True, timeout=900)
        assert r.returncode == 0, r.stderr[-2000:]
        d = json.loads([l for l in r.stdout.splitlines() if l.strip()][-1])
        assert d["steps"] == 20 and d["warmup"] == 5
        over.append(20 * d["ms_per_step"] * (1.0 - d["value"] / d["value_steady_state"]))
        if over[-1] < 4.0:
            break
    assert min(over) < 4.0, (over, d["value"], d["value_steady_state"])


@pytest.mark.parametrize("dest", ["shm", "file", "host"])
def test_two_rank_control_flow_on_one_gpu(dest, tmp_path):
    """The N > 1 path of bench.py rehearsed with two ranks that share the one GPU (gloo instead of RCCL, which cannot
    form a group of ranks on the same device): count exchange, gather, and image 0 of BOTH ranks compared with the
    oracle out of what rank 0 holds at the end of a step -- for each of the three places the other rank's lists can
    reach rank 0's host memory through:
      shm   rank 0 reads them in place from the node-shared result buffers in /dev/shm (sized by need);
      file  the same with the buffers in files (what the library does when /dev/shm has no room: forced here);
      host  no shared buffers at all (a name the library refuses): rank 0 copies the gathered lists out of its HBM."""
    import socket
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    env = dict(os.environ, HESS_BENCH_BACKEND="gloo", HESS_BENCH_SAME_GPU="1")
    if dest == "file":   # (HESS_SHARE_FORCE_FILE is a test hook of the developer build: the ranks load that library)
        import hessgpu_amd
        env.update(HESS_SHARE_FORCE_FILE="1", HESS_SHARE_DIR=str(tmp_path), HESS_LIB=hessgpu_amd.DEV_LIB_PATH)
    if dest == "host":
        env["HESS_BENCH_SHM_PREFIX"] = "no/such"
    bench = [os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "4", "--warmup", "1", "--contexts", "2", "--batch", "3", "--no-profile"]
    if dest == "shm":   # the BARE command, as the driver runs N = 1: bench.py starts its own two ranks (hessgpu_amd/launch.py)
        cmd = [sys.executable] + bench
        env = {k: v for k, v in env.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    else:               # under the launcher the contract names
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
               "--master-addr", "127.0.0.1", "--master-port", str(port)] + bench
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.strip().startswith("{")]
    assert len(lines) == 1, r.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["scaling"] == "weak"
    assert d["config"]["gather_dest"] == ("host" if dest == "host" else "shm")
    assert d["parity_checked"] is True and d["parity_checked_ranks"] == 2
    assert abs(d["value"] - 2 * 3 * 4 * 1920 * 1080 / (d["ms_per_step"] * 4 * 1e-3) / 1e6) / d["value"] < 0.01
    if dest != "host":
        mb, in_files = d["config"]["shared_result_buffers_mb"], d["config"]["shared_result_buffers_in_files_mb"]
        # sized by need: 2 ranks x 2 contexts x 3 images x ~5.6 k features x 536 B + 25 % -- far below the worst case
        # (3 x 16384 records per context: 105 MB for the four)
        assert 30 < mb < 70, mb
        assert (in_files == mb) if dest == "file" else (in_files == 0)
        assert "node-shared result buffers:" in r.stderr
    else:
        assert "gather destination of every rank: host" in r.stderr
    assert not [f for f in os.listdir("/dev/shm") if f.startswith("hessbench_")]
    assert not list(tmp_path.iterdir())          # the files are gone with their contexts


def _bare_env(**extra):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR")}
    env.update(extra)
    return env


def test_a_rank_that_dies_mid_run_ends_the_self_launched_job():
    """Rank 1 of a self-launched two-rank job leaves with code 17 before its third timed step; rank 0 is then blocked in
    the exchange of that step.  The parent ends it and reports 17 -- within seconds, not after the backend's timeout."""
    import time
    t0 = time.monotonic()
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "6", "--warmup", "1",
                        "--contexts", "2", "--batch", "2", "--no-profile", "--no-cpu-baseline"], capture_output=True, text=True, timeout=600,
                       env=_bare_env(HESS_BENCH_BACKEND="gloo", HESS_BENCH_SAME_GPU="1", HESS_BENCH_TEST_EXIT="1:2:17"))
    assert r.returncode == 17, (r.returncode, r.stderr[-2000:])
    assert "rank(s) ended non-zero" in r.stderr and "(1, 17)" in r.stderr   # (rank 0 may fail on the broken connection in the same poll)
    assert not [l for l in r.stdout.splitlines() if l.strip().startswith("{")]      # no result line from a broken job
    assert time.monotonic() - t0 < 300


def test_one_rank_through_the_launcher_agrees_with_the_plain_run():
    """--gpus 1 --spawn: one child rank on the N > 1 code path (RCCL group of one, exchange on the helper thread).  Its
    value agrees with the plain in-process N = 1 run: taking the exchange off the submitting thread costs nothing."""
    common = ["--steps", "60", "--warmup", "10", "--no-configs4", "--no-api-leg", "--no-host-leg", "--no-cpu-baseline", "--no-profile", "--no-steady"]
    vals = {}
    for name, extra in (("plain", []), ("spawn", ["--spawn"]), ("plain2", []), ("spawn2", ["--spawn"])):
        r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1"] + common + extra,
                           capture_output=True, text=True, timeout=600, env=_bare_env())
        assert r.returncode == 0, r.stderr[-2000:]
        d = json.loads([l for l in r.stdout.splitlines() if l.strip().startswith("{")][-1])
        assert d["n_gpus"] == 1
        vals[name] = d["value"]
        if extra:
            assert "RCCL" in d["config"]["sharding"] and d["config"]["exchange"].startswith("helper thread")
    plain, spawn = max(vals["plain"], vals["plain2"]), max(vals["spawn"], vals["spawn2"])
    # (the dist path uses 6 contexts like the plain run here: FORCE_DIST keeps world == 1)
    # (best of two each; identical runs on one box differ by up to 7 % -- the same-call A/B lines under profiles/ -- so this
    # bounds a systematic cost of the N > 1 path, it does not resolve 2 %: measured 0 - 3 %)
    assert spawn > 0.92 * plain, vals
