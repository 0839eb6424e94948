#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}
cd $R; mkdir -p gpurun_out/r03short
run() {
  timeout -k 10 200 python bench.py --batch 8 --steps $1 --warmup $2 --contexts 6 --no-cpu-baseline --no-api-leg --no-configs4 --no-host-leg --no-profile > gpurun_out/r03short/y.json 2>/dev/null
  python - <<PY
import json
d=json.loads(open("gpurun_out/r03short/y.json").read().strip().splitlines()[-1]); print("steps $1 warmup $2:", d["value"], "Mpix/s,", round(d["ms_per_step"]*$1,2), "ms for the region")
PY
}
for rep in 1 2; do
run 20 5; run 20 50; run 20 200; run 40 5; run 40 200; run 200 10
done
