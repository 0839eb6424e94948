#!/usr/bin/env python3
"""Effective shader clock per kernel from a `--pmc GRBM_GUI_ACTIVE --kernel-trace` pass (tools/profile_round.sh step 7):
GRBM_GUI_ACTIVE is summed over the 8 XCDs, so clock = counter / 8 / duration (MI355X_MICROARCH.md, "DVFS give-back";
reads high on dispatches shorter than about 0.3 ms).   python tools/profile_clock_summary.py gpurun_out/<tag> > clock.csv"""
import collections
import csv
import glob
import re
import sys


def main():
    cc = glob.glob(sys.argv[1] + "/pmc_clk/**/*counter_collection.csv", recursive=True)[0]
    agg = collections.defaultdict(lambda: [0.0, 0.0, 0])
    for r in csv.DictReader(open(cc)):
        if r["Counter_Name"] != "GRBM_GUI_ACTIVE":
            continue
        name = re.sub(r"\(.*", "", r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").replace("hess::", ""))
        a = agg[name]
        a[0] += float(r["Counter_Value"]); a[1] += float(r["End_Timestamp"]) - float(r["Start_Timestamp"]); a[2] += 1
    w = csv.writer(sys.stdout)   # (kernel names carry commas: quoted)
    w.writerow(["kernel", "launches", "avg_us", "gui_active_per_launch", "effective_clock_ghz"])
    for n, (v, d, k) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        w.writerow([n, k, f"{d / k / 1e3:.2f}", f"{v / k:.4g}", f"{(v / 8.0) / d:.3f}"])


if __name__ == "__main__":
    main()
