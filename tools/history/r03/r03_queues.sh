#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/r03
mkdir -p $OUT; cd $R
for rep in 1 2; do
for q in 4 6 8; do
for c in 6 8; do
  GPU_MAX_HW_QUEUES=$q timeout -k 10 300 python bench.py --contexts $c --no-cpu-baseline --no-api-leg --no-configs4 --no-profile > $OUT/q_${q}_${c}_$rep.json 2> $OUT/q.err
  python - <<PY
import json
d=json.load(open("$OUT/q_${q}_${c}_$rep.json")); print("hw queues $q contexts $c:", d["value"], d["value_host_to_host"], d["latency_ms_single_image"])
PY
done
done
done
