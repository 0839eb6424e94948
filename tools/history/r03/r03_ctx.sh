#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/r03ctx
mkdir -p $OUT; cd $R
for rep in 1 2; do
for c in 4 5 6 7 8; do
  timeout -k 10 300 python bench.py --steps 150 --contexts $c --no-cpu-baseline --no-api-leg --no-configs4 --no-host-leg --no-profile > $OUT/c${c}_$rep.json 2>/dev/null
  python - <<PY
import json
b=json.loads(open("$OUT/c${c}_$rep.json").read().strip().splitlines()[-1])
print("contexts $c", b["value"], b["ms_per_step"])
PY
done
done
