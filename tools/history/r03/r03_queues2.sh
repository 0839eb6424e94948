#!/bin/bash
# GPU_MAX_HW_QUEUES below the default of 4, six contexts
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/r03q2
mkdir -p $OUT; cd $R
run() {
  local name=$1; shift
  env "$@" timeout -k 10 200 python bench.py --steps 100 --no-cpu-baseline --no-api-leg --no-configs4 --no-host-leg ${CTX} > $OUT/$name.json 2> $OUT/$name.err || { echo "$name failed"; tail -3 $OUT/$name.err; return 1; }
  python - $name $OUT/$name.json <<'PY'
import json,sys
d=json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
print(sys.argv[1], d["value"], "ms/step", d["ms_per_step"])
PY
}
run q4 A=1 && run q3 GPU_MAX_HW_QUEUES=3 && run q2 GPU_MAX_HW_QUEUES=2 && run q1 GPU_MAX_HW_QUEUES=1 && run q5 GPU_MAX_HW_QUEUES=5 &&
CTX="--contexts 4" run q2c4 GPU_MAX_HW_QUEUES=2 && CTX="--contexts 8" run q3c8 GPU_MAX_HW_QUEUES=3 && CTX="--contexts 3" run q3c3 GPU_MAX_HW_QUEUES=3 && run q4b A=1
