#!/bin/bash
# descriptors of a batch in two launches, the first half's results crossing while the second is computed (default)
# against one launch and one transfer: HESS_DESC_PARTS = 1, 2, 4
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/r03split
mkdir -p $OUT; cd $R
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_reference_inputs_gpu.py tests/test_shared_results.py -m gpu -x -q > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -2 $OUT/pytest.log
for rep in 1 2; do
for ns in 1 2 4; do
  export HESS_DESC_PARTS=$ns
  timeout -k 10 300 python tools/bench_host_path.py > $OUT/hp_$ns.json 2>/dev/null
  timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-api-leg --no-configs4 > $OUT/b20_$ns.json 2>/dev/null
  timeout -k 10 300 python bench.py --steps 200 --warmup 10 --no-cpu-baseline --no-api-leg --no-configs4 --no-host-leg --no-profile > $OUT/b200_$ns.json 2>/dev/null
  python - <<PY
import json
h=json.load(open("$OUT/hp_$ns.json")); b=json.loads(open("$OUT/b20_$ns.json").read().strip().splitlines()[-1]); b2=json.loads(open("$OUT/b200_$ns.json").read().strip().splitlines()[-1])
print("parts=$ns", "batch8 dev/pinned/pageable ms:", h["batch_8"]["device_resident"]["ms_per_batch"], h["batch_8"]["host_pinned"]["ms_per_batch"], h["batch_8"]["host_pageable"]["ms_per_batch"], "batch16 dev:", h["batch_16"]["device_resident"]["ms_per_batch"], "| K=20:", b["value"], "steady", b.get("value_steady_state"), "h2h", b["value_host_to_host"], "| K=200:", b2["value"])
PY
done
done
