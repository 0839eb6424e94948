#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/r03
mkdir -p $OUT; cd $R
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $OUT/pytest_f.log 2>&1; echo "pytest rc=$?"; tail -3 $OUT/pytest_f.log
for rep in 1 2; do
for np in 0 1; do
  if [ $np = 1 ]; then export HESS_NO_PAIR=1; else unset HESS_NO_PAIR; fi
  timeout -k 10 300 python tools/bench_host_path.py > $OUT/pair_hp_${np}_$rep.json 2>/dev/null
  timeout -k 10 300 python bench.py --no-cpu-baseline --no-api-leg --no-configs4 --no-host-leg > $OUT/pair_b_${np}_$rep.json 2>/dev/null
  python - <<PY
import json
h=json.load(open("$OUT/pair_hp_${np}_$rep.json")); b=json.load(open("$OUT/pair_b_${np}_$rep.json"))
print("no_pair=$np", "batch1 dev/pageable ms:", h["batch_1"]["device_resident"]["ms_per_batch"], h["batch_1"]["host_pageable"]["ms_per_batch"], "batch8 dev ms:", h["batch_8"]["device_resident"]["ms_per_batch"], "| value", b["value"], "gauss ms/step", b["kernel_ms_per_step"]["gauss"])
PY
done
done
