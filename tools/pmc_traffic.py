#!/usr/bin/env python3
"""Summarise two rocprofv3 PMC passes (one with --pmc FETCH_SIZE, one with --pmc WRITE_SIZE) into
profiles/gauss_traffic.json: HBM bytes per launch of the Gaussian kernel (and of every other kernel).

  python tools/pmc_traffic.py <dir of the FETCH_SIZE pass> <dir of the WRITE_SIZE pass> <algorithmic bytes/launch> > profiles/gauss_traffic.json

gfx950 correction (MI355X_MICROARCH.md, HBM / rocprofv3 section): FETCH_SIZE and WRITE_SIZE are in KiB and
FETCH_SIZE counts half the bytes of wide coalesced reads: bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024.
"""
import collections
import csv
import glob
import json
import re
import sys


def short(name):
    name = name.replace("(anonymous namespace)::", "").replace("void ", "").replace("hess::", "")
    return re.sub(r"\(.*", "", name)


def load(d, counter):
    f = glob.glob(f"{d}/**/*counter_collection.csv", recursive=True)[0]
    tot, launches = collections.Counter(), collections.defaultdict(set)
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] != counter:
            continue
        n = short(r["Kernel_Name"])
        tot[n] += float(r["Counter_Value"])
        launches[n].add(r["Dispatch_Id"])
    return tot, {k: len(v) for k, v in launches.items()}


def main():
    fdir, wdir = sys.argv[1], sys.argv[2]
    algo = float(sys.argv[3]) if len(sys.argv) > 3 else None
    fetch, nl = load(fdir, "FETCH_SIZE")
    write, _ = load(wdir, "WRITE_SIZE")
    per = []
    gb, gl = 0.0, 0
    for k in sorted(fetch):
        b = (2.0 * fetch[k] + write.get(k, 0.0)) * 1024.0
        per.append({"kernel": k, "launches": nl[k], "FETCH_SIZE_KB_sum": fetch[k], "WRITE_SIZE_KB_sum": write.get(k, 0.0),
                    "hbm_bytes_per_launch_corrected": b / nl[k]})
        if k.startswith(("gauss_kernel", "gauss_pair_kernel", "gauss_top_kernel", "gauss_first_kernel")):
            gb += b
            gl += nl[k]
    out = {
        "command": "rocprofv3 --kernel-trace --pmc FETCH_SIZE (and, separately, --pmc WRITE_SIZE) -- python3 bench.py "
                   "--steps 2 --warmup 1 --no-cpu-baseline --no-profile --contexts 1",
        "correction": "gfx950: FETCH_SIZE counts half the bytes of wide coalesced reads (MI355X_MICROARCH.md, HBM): "
                      "bytes = (2*FETCH_SIZE + WRITE_SIZE) * 1024",
        "kernel": "gauss_kernel, gauss_pair_kernel, gauss_top_kernel, gauss_first_kernel (all instantiations)",
        "launches": gl,
        "hbm_bytes_per_launch": gb / max(gl, 1),
        "algorithmic_bytes_per_launch": algo,
        "per_kernel": per,
    }
    json.dump(out, sys.stdout, indent=1)


if __name__ == "__main__":
    main()
