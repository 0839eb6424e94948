#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}
cd $R; mkdir -p gpurun_out/r03steps
for rep in 1 2; do
for s in "20 5" "20 2" "50 5" "200 10"; do
  set -- $s
  timeout -k 10 200 python bench.py --steps $1 --warmup $2 --no-cpu-baseline --no-api-leg --no-configs4 --no-host-leg --no-profile > gpurun_out/r03steps/s$1_$2.json 2>/dev/null
  python - <<PY
import json
d=json.loads(open("gpurun_out/r03steps/s$1_$2.json").read().strip().splitlines()[-1]); print("steps $1 warmup $2:", d["value"], d["ms_per_step"])
PY
done
done
