"""The committed PMC / trace summaries that bench.py copies numbers from (profiles/*.json) are plausible and agree with the
committed kernel table they were distilled from.  Round 3 published a vector-issue fraction of 15.94 because one of these
files was wrong by a factor of 68 and nothing looked at it without a GPU."""
import csv
import json
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
P = os.path.join(ROOT, "profiles")


def _j(name):
    with open(os.path.join(P, name)) as f:
        return json.load(f)


def test_gauss_traffic_is_plausible():
    t = _j("gauss_traffic.json")
    # a 1080p pyramid with its fused det-H / gradient stage: 1.5e7 - 2.5e7 vector instructions per image
    assert 1.0e7 < t["valu_insts_per_image"] < 3.0e7, t["valu_insts_per_image"]
    assert t["valu_insts_steps_in_pass"] >= 1
    # HBM traffic per launch within a few per cent of the bytes the launches have to move (each level they store read once,
    # written once); the algorithmic figure (the reference's layout) also counts the two levels kept in LDS
    moved = t.get("bytes_moved_per_launch", t["algorithmic_bytes_per_launch"])
    assert 0.95 < t["hbm_bytes_per_launch"] / moved < 1.15
    assert 1.0 <= t["algorithmic_bytes_per_launch"] / moved < 1.3
    # the nominal issue peak cannot be exceeded by the rate this implies at the committed kernel time (about 0.5 ms per
    # step of eight images): instructions x 8 / 0.5 ms < 1228.8 Ginst/s
    assert t["valu_insts_per_image"] * 8 / 0.5e-3 / 1e9 < 1228.8


def test_descriptor_counters_are_plausible():
    d = _j("descriptor_counters.json")
    # (round 4's cell-by-cell kernel: 5.6e3; the pixel raster of round 5: 2.7e3)
    assert 1.5e3 < d["valu_insts_per_feature"] < 1.2e4
    assert 0.0 < d["wave_time_issuing_valu"] < 1.0 and 0.0 < d["wave_time_waiting"] < 1.0
    assert d["wave_time_issuing_valu"] + d["wave_time_waiting"] + d["wave_time_issue_stalled"] <= 1.0
    assert 0.3 < d["hbm_bytes_per_launch"] / d["algorithmic_bytes_per_launch"] < 2.0
    assert abs(d["valu_insts_per_launch"] / d["features_per_launch"] - d["valu_insts_per_feature"]) < 1.0


def test_kernel_stats_top_is_the_top_row_of_the_committed_table():
    top = _j("kernel_stats_top.json")
    assert 0.0 < top["frac"] <= 1.0 and abs(top["frac"] - top["achieved"] / top["peak"]) < 1e-3
    m = re.search(r"profiles/(\S+_kernel_stats_contexts1\.csv)", top["source"])
    assert m, top["source"]
    rows = [r for r in csv.DictReader(open(os.path.join(P, m.group(1)))) if "hess::" in r["Name"]]
    rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
    name = re.sub(r"\(.*", "", rows[0]["Name"].replace("(anonymous namespace)::", "").replace("void ", "").replace("hess::", ""))
    # (an instantiation of the Gaussian kernel at the top stands for the family's entry, which names it)
    assert top["kernel"].startswith(name.split("<")[0]) or (name.startswith("gauss") and top["kernel"].startswith("gauss") and name in top["kernel"])
    if top["kernel"] == name:
        assert abs(top["avg_launch_us"] - float(rows[0]["AverageNs"]) / 1e3) < 0.05
        assert abs(top["achieved"] - top["algorithmic_bytes_per_launch"] / (top["avg_launch_us"] * 1e-6) / 1e9) < 1.0


def test_chip_constants_used_by_the_bench_line():
    assert 800.0 < _j("valu_peak.json")["sustained_fma_ginst_all_cus"] <= 1228.8
    assert 3000.0 < _j("hbm_mix.json")["one_read_four_writes_gbs"] < 8000.0


def test_committed_summaries_say_what_they_were_measured_on_and_bench_drops_stale_ones(monkeypatch):
    """Every kernel-profile summary bench.py copies numbers from records the hash of the kernel sources it was measured on
    (hessgpu_amd/build.py sources_digest()) and the commit; bench.py's reader marks one measured on OTHER sources as stale and
    hands out none of its numbers (round 5: an intermediate run multiplied a committed instruction count by a new kernel's
    time and printed a fraction of 1.10)."""
    import importlib.util
    import sys

    for name in ("gauss_traffic.json", "descriptor_counters.json", "kernel_stats_top.json"):
        d = _j(name)
        assert re.fullmatch(r"[0-9a-f]{16}", d["kernel_sources_sha16"]) and d["commit"], name
    sys.path.insert(0, ROOT)
    spec = importlib.util.spec_from_file_location("bench_for_test", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    from hessgpu_amd import build

    assert re.fullmatch(r"[0-9a-f]{16}", build.sources_digest())
    # as committed: current or stale, but consistently so
    d = bench._profile_json("descriptor_counters.json")
    stale = bool(d.get("stale"))
    assert (bench._profile_value("descriptor_counters.json", "valu_insts_per_feature") is None) == stale
    # a summary of other sources: stale, no numbers
    bench._sources_sha16[:] = ["0" * 16]
    d = bench._profile_json("descriptor_counters.json")
    assert d["stale"] is True and "0000000000000000" in d["stale_reason"]
    assert bench._profile_value("descriptor_counters.json", "valu_insts_per_feature") is None
    # chip constants carry no source hash: always used
    assert bench._profile_value("valu_peak.json", "sustained_fma_ginst_all_cus") > 0
