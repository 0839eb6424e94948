#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/r03
mkdir -p $OUT; cd $R
for rep in 1 2; do
for m in mirror dma; do
for c in 3 4 5 6; do
  HESS_DELIVERY=$m timeout -k 10 300 python bench.py --contexts $c --no-cpu-baseline --no-api-leg --no-configs4 --no-profile > $OUT/ctx_${m}_${c}_$rep.json 2> $OUT/ctx.err
  python - <<PY
import json
d=json.load(open("$OUT/ctx_${m}_${c}_$rep.json")); print("$m contexts $c:", d["value"], d["value_host_to_host"])
PY
done
done
done
