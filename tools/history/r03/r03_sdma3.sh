#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/r03
mkdir -p $OUT; cd $R
for rep in 1 2 3; do
for cfg in "mirror 3" "dma 5" "dma 6" "dma 8" "mirror 2"; do
  set -- $cfg
  HESS_DELIVERY=$1 timeout -k 10 300 python bench.py --contexts $2 --steps 200 --warmup 10 --no-cpu-baseline --no-api-leg --no-configs4 --no-profile > $OUT/ctxl_$1_$2_$rep.json 2> $OUT/ctx.err
  python - <<PY
import json
d=json.load(open("$OUT/ctxl_$1_$2_$rep.json")); print("$1 contexts $2:", d["value"], d["value_host_to_host"])
PY
done
done
