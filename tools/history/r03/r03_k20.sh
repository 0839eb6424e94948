#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}
cd $R; mkdir -p gpurun_out/r03k20
run() {
  local c=$1 d=$2
  HESS_DELIVERY=$d timeout -k 10 200 python bench.py --steps 20 --warmup 5 --contexts $c --no-cpu-baseline --no-api-leg --no-configs4 --no-host-leg --no-profile > gpurun_out/r03k20/z.json 2>/dev/null
  python - <<PY
import json
d=json.loads(open("gpurun_out/r03k20/z.json").read().strip().splitlines()[-1]); print("contexts $c delivery $d:", d["value"], d["ms_per_step"])
PY
}
for rep in 1 2; do
for c in 3 4 5 6 7; do run $c dma; done
for c in 3 4 6; do run $c mirror; done
done
