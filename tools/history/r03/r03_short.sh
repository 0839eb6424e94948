#!/bin/bash
# 160 timed images (what the driver's --steps 20 times at batches of 8) as units of 8, 4, 2 images
R=${GRAFT_REPO_ROOT:-$PWD}
cd $R; mkdir -p gpurun_out/r03short
run() {
  timeout -k 10 200 python bench.py --batch $1 --steps $2 --warmup $3 --contexts $4 --no-cpu-baseline --no-api-leg --no-configs4 --no-host-leg --no-profile > gpurun_out/r03short/x.json 2>/dev/null
  python - <<PY
import json
d=json.loads(open("gpurun_out/r03short/x.json").read().strip().splitlines()[-1]); print("batch $1 steps $2 warmup $3 contexts $4:", d["value"], "Mpix/s,", round(d["ms_per_step"]*$2,2), "ms for the region")
PY
}
for rep in 1 2; do
run 8 20 5 6; run 8 20 5 3; run 4 40 10 6; run 4 40 10 8; run 4 40 10 4; run 2 80 20 8; run 2 80 20 6; run 2 80 20 12; run 1 160 40 8
done
