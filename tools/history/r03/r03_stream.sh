#!/bin/bash
# streaming det-H / gradient stores (default) against plain stores (variant "plain"): single image, batch 8/16, configs[4]
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/r03st
mkdir -p $OUT; cd $R
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -2 $OUT/pytest.log
for rep in 1 2; do
for v in cur plain; do
  if [ $v = cur ]; then unset HESS_LIB; else export HESS_LIB=$R/tools/_variants/$v/libhessgpu.so; fi
  timeout -k 10 300 python tools/bench_host_path.py > $OUT/hp_${v}_$rep.json 2>/dev/null
  timeout -k 10 300 python bench.py --steps 150 --no-cpu-baseline --no-api-leg > $OUT/b_${v}_$rep.json 2>/dev/null
  timeout -k 10 300 python bench.py --steps 100 --batch 16 --no-cpu-baseline --no-api-leg --no-configs4 --no-host-leg > $OUT/b16_${v}_$rep.json 2>/dev/null
  python - <<PY
import json
h=json.load(open("$OUT/hp_${v}_$rep.json")); b=json.loads(open("$OUT/b_${v}_$rep.json").read().strip().splitlines()[-1]); b16=json.loads(open("$OUT/b16_${v}_$rep.json").read().strip().splitlines()[-1])
print("$v", "batch1 dev/pinned/pageable ms:", h["batch_1"]["device_resident"]["ms_per_batch"], h["batch_1"]["host_pinned"]["ms_per_batch"], h["batch_1"]["host_pageable"]["ms_per_batch"], "| value", b["value"], "h2h", b["value_host_to_host"], "lat", b["latency_ms_single_image"], "cfg4", b["configs4"]["Mpix_per_s_one_context"], b["configs4"]["Mpix_per_s_three_contexts"], "| b16", b16["value"], {k:round(x,3) for k,x in b["kernel_ms_per_step"].items()})
PY
done
done
