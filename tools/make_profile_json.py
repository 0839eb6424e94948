#!/usr/bin/env python3
"""Distils one tools/profile_round.sh output directory into the two small JSON files bench.py reads:

  python tools/make_profile_json.py gpurun_out/<tag> <tag>    ->  profiles/gauss_traffic.json, profiles/descriptor_counters.json

gauss_traffic.json        HBM bytes per launch of the Gaussian kernel (tools/pmc_traffic.py, gfx950 correction applied)
descriptor_counters.json  descriptor_kernel: vector instructions per launch and per feature (SQ_INSTS_VALU), LDS instructions,
                          bank-conflict share, wavefront-time split, effective clock, HBM bytes per launch
"""
import csv
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    d, tag = sys.argv[1], sys.argv[2]
    bench = json.loads(open(os.path.join(d, "bench_ctx1.json")).read().strip().splitlines()[-1])
    traffic = json.load(open(os.path.join(d, "traffic.json")))
    rows = {r["kernel"]: r for r in csv.DictReader(open(os.path.join(d, "counters.csv")))}
    clock = {r["kernel"]: r for r in csv.DictReader(open(os.path.join(d, "clock.csv")))} if os.path.exists(os.path.join(d, "clock.csv")) else {}
    g = bench["roofline"] if bench["roofline"]["kernel"].startswith("gauss") else bench["roofline_secondary"]
    traffic["algorithmic_bytes_per_launch"] = g["algorithmic_bytes_per_launch"]
    traffic["source"] = f"profiles/{tag}_* (tools/profile_round.sh {tag})"
    # vector instructions of all Gaussian launches per image: the SQ pass runs 3 steps (--steps 2 --warmup 1) of one batch
    batch = bench["config"]["images_per_gpu_per_step"]
    gv = sum(float(r["SQ_INSTS_VALU"]) * int(r["launches"]) for k, r in rows.items() if k.startswith(("gauss_kernel", "gauss_pair_kernel")))
    traffic["valu_insts_per_image"] = round(gv / (3 * batch), 1)
    json.dump(traffic, open(os.path.join(ROOT, "profiles", "gauss_traffic.json"), "w"), indent=1)
    dd = bench["roofline"] if bench["roofline"]["kernel"].startswith("descriptor") else bench["roofline_secondary"]
    dk = dd["kernel"].split()[0]  # the form the benched batch size uses: descriptor_kernel<false> (copier delivery) or <true> (host mirror)
    r = rows[dk]
    feats = dd["features_per_launch"]
    f = lambda k: float(r[k])
    tr = next((p for p in traffic["per_kernel"] if p["kernel"] == dk), None)
    out = {
        "source": f"profiles/{tag}_counters.csv, {tag}_clock.csv (tools/profile_round.sh {tag}: bench.py --contexts 1, batch {bench['config']['images_per_gpu_per_step']})",
        "features_per_launch": feats,
        "valu_insts_per_launch": f("SQ_INSTS_VALU"),
        "valu_insts_per_feature": round(f("SQ_INSTS_VALU") / feats, 1),
        "lds_insts_per_feature": round(f("SQ_INSTS_LDS") / feats, 1),
        "lds_bank_conflict_share_of_lds_cycles": round(f("SQ_LDS_BANK_CONFLICT") / f("SQ_LDS_IDX_ACTIVE"), 3),
        "wave_time_waiting": round(f("SQ_WAIT_ANY") / f("SQ_WAVE_CYCLES"), 3),
        "wave_time_issue_stalled": round(f("SQ_WAIT_INST_ANY") / f("SQ_WAVE_CYCLES"), 3),
        "wave_time_issuing_valu": round(f("SQ_ACTIVE_INST_VALU") / f("SQ_WAVE_CYCLES"), 3),
        "effective_clock_ghz": float(clock[dk]["effective_clock_ghz"]) if dk in clock else None,
        "hbm_bytes_per_launch": tr["hbm_bytes_per_launch_corrected"] if tr else None,
        "algorithmic_bytes_per_launch": dd["algorithmic_bytes_per_launch"],
    }
    json.dump(out, open(os.path.join(ROOT, "profiles", "descriptor_counters.json"), "w"), indent=1)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
