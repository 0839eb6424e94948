#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/r03
mkdir -p $OUT
cd $R
for k in 0 2 3 4 0 2 3; do
  HESS_DESC_WG_PER_CU=$k HESS_DELIVERY=mirror timeout -k 10 300 python bench.py --no-cpu-baseline --no-api-leg --no-configs4 --no-host-leg > $OUT/descgrid_${k}_$(date +%s%N).json 2> $OUT/descgrid.err
  python - <<PY
import json,glob
f=sorted(glob.glob("$OUT/descgrid_${k}_*.json"))[-1]
d=json.load(open(f)); print("wg_per_cu=$k mirror", d["value"], d["kernel_ms_per_step"])
PY
done
for k in 0 2 3; do
  HESS_DESC_WG_PER_CU=$k HESS_DELIVERY=blit timeout -k 10 300 python bench.py --no-cpu-baseline --no-api-leg --no-configs4 --no-host-leg > $OUT/descgridb_${k}_$(date +%s%N).json 2> $OUT/descgrid.err
  python - <<PY
import json,glob
f=sorted(glob.glob("$OUT/descgridb_${k}_*.json"))[-1]
d=json.load(open(f)); print("wg_per_cu=$k blit", d["value"], d["kernel_ms_per_step"])
PY
done
