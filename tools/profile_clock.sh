#!/bin/bash
# Effective shader clock per kernel: GRBM_GUI_ACTIVE (summed over the 8 XCDs) / 8 / kernel duration (MI355X_MICROARCH.md,
# "DVFS give-back").  gpurun -- 'bash tools/profile_clock.sh TAG [bench args]'  -> gpurun_out/TAG/clock.csv
set -e
TAG=${1:-clk}; shift || true
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_clk -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-profile --no-host-leg --contexts 1 "$@" > /dev/null 2> $OUT/pmc_clk.err
python3 - $OUT <<'PY'
import csv, glob, sys, collections, re
out = sys.argv[1]
cc = glob.glob(out + "/pmc_clk/**/*counter_collection.csv", recursive=True)[0]
rows = list(csv.DictReader(open(cc)))
agg = collections.defaultdict(lambda: [0.0, 0.0, 0])
for r in rows:
    if r["Counter_Name"] != "GRBM_GUI_ACTIVE":
        continue
    name = re.sub(r"\(.*", "", r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").replace("hess::", ""))
    dur = float(r["End_Timestamp"]) - float(r["Start_Timestamp"]) if "End_Timestamp" in r else 0.0
    a = agg[name]; a[0] += float(r["Counter_Value"]); a[1] += dur; a[2] += 1
with open(out + "/clock.csv", "w") as f:
    f.write("kernel,launches,avg_us,gui_active_per_launch,effective_clock_ghz\n")
    for n, (v, d, k) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        f.write(f"{n},{k},{d / k / 1e3 if d else 0:.2f},{v / k:.4g},{(v / 8.0) / d if d else 0:.3f}\n")
print(open(out + "/clock.csv").read())
PY
