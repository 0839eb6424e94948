#!/usr/bin/env python3
"""Per-kernel summary of one rocprofv3 SQ counter pass (tools/profile_round.sh, step 5):
fractions of wavefront time parked (SQ_WAIT_ANY), issue-stalled (SQ_WAIT_INST_ANY) and issuing
(SQ_ACTIVE_INST_ANY; VALU and LDS parts), and VALU instructions per launch.

  python tools/pmc_sq_summary.py gpurun_out/<tag>/pmc_sq > profiles/<round>_sq_summary.csv
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
    f = glob.glob(f"{sys.argv[1]}/**/*counter_collection.csv", recursive=True)[0]
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    launches = collections.defaultdict(set)
    for r in csv.DictReader(open(f)):
        n = short(r["Kernel_Name"])
        agg[n][r["Counter_Name"]] += float(r["Counter_Value"])
        launches[n].add(r["Dispatch_Id"])
    w = csv.writer(sys.stdout)
    w.writerow(["kernel", "launches", "wave_cycles", "wait_any_frac", "wait_inst_frac", "active_inst_frac",
                "active_valu_frac", "active_lds_frac", "valu_insts_per_launch"])
    for n in sorted(agg, key=lambda k: -agg[k]["SQ_WAVE_CYCLES"]):
        x = agg[n]
        wc = x["SQ_WAVE_CYCLES"] or 1.0
        w.writerow([n, len(launches[n]), f"{wc:.4g}", f"{x['SQ_WAIT_ANY'] / wc:.3f}", f"{x['SQ_WAIT_INST_ANY'] / wc:.3f}",
                    f"{x['SQ_ACTIVE_INST_ANY'] / wc:.3f}", f"{x['SQ_ACTIVE_INST_VALU'] / wc:.3f}",
                    f"{x['SQ_ACTIVE_INST_LDS'] / wc:.3f}", f"{x['SQ_INSTS_VALU'] / len(launches[n]):.4g}"])


if __name__ == "__main__":
    main()
