#!/usr/bin/env python3
"""Per-kernel table of raw rocprofv3 PMC counters, averaged per launch, from one or more counter passes.

  python tools/pmc_counters.py gpurun_out/<tag>/pmc_sq gpurun_out/<tag>/pmc_sq2 [--kernels descriptor gauss] > profiles/<name>.csv

Every *counter_collection.csv below the given directories is read; counters are summed over the dispatch's
dimensions (XCDs/SEs) and averaged over the launches of a kernel.  Columns appear in the order counters are met.
Derived columns (when their inputs exist):
  valu_issue_frac   SQ_INSTS_VALU / launch duration-free estimate is not possible here; instead
  valu_per_wave     SQ_INSTS_VALU / SQ_WAVES,   lds_per_wave  SQ_INSTS_LDS / SQ_WAVES
  busy_cycles       SQ_BUSY_CYCLES per launch (quad-cycles summed over SEs)
"""
import collections
import csv
import glob
import re
import sys


def short(name):
    name = name.replace("(anonymous namespace)::", "").replace("void ", "").replace("hess::", "")
    return re.sub(r"\(.*", "", name)


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    want = None
    if "--kernels" in sys.argv:
        i = sys.argv.index("--kernels")
        want = sys.argv[i + 1:]
        args = [a for a in args if a not in want]
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    launches = collections.defaultdict(lambda: collections.defaultdict(set))
    order = []
    for d in args:
        for f in glob.glob(f"{d}/**/*counter_collection.csv", recursive=True):
            for r in csv.DictReader(open(f)):
                n = short(r["Kernel_Name"])
                if want and not any(w in n for w in want):
                    continue
                c = r["Counter_Name"]
                if c not in order:
                    order.append(c)
                agg[n][c] += float(r["Counter_Value"])
                launches[n][c].add((f, r["Dispatch_Id"]))
    w = csv.writer(sys.stdout)
    w.writerow(["kernel", "launches"] + order + ["valu_per_wave", "lds_per_wave"])
    for n in sorted(agg, key=lambda k: -agg[k].get("SQ_WAVE_CYCLES", 0.0)):
        x = agg[n]
        per = {c: x[c] / max(1, len(launches[n][c])) for c in order if c in x}
        nl = min(len(v) for v in launches[n].values())  # dispatches of ONE pass (SQ_WAVES is collected in both)
        waves = per.get("SQ_WAVES", 0.0)
        row = [n, nl] + [f"{per.get(c, float('nan')):.5g}" for c in order]
        row += [f"{per.get('SQ_INSTS_VALU', 0) / waves:.5g}" if waves else "", f"{per.get('SQ_INSTS_LDS', 0) / waves:.5g}" if waves else ""]
        w.writerow(row)


if __name__ == "__main__":
    main()
