#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/r03
mkdir -p $OUT; cd $R
A="--steps 600 --warmup 20 --no-cpu-baseline --no-api-leg --no-configs4 --no-profile --no-host-leg"
python bench.py $A > $OUT/two_single.json 2>/dev/null
python bench.py --contexts 3 $A > $OUT/two_a.json 2>/dev/null &
python bench.py --contexts 3 $A > $OUT/two_b.json 2>/dev/null &
wait
python bench.py --contexts 6 $A > $OUT/two_c.json 2>/dev/null &
python bench.py --contexts 6 $A > $OUT/two_d.json 2>/dev/null &
wait
python - <<PY
import json
for n in ("two_single","two_a","two_b","two_c","two_d"):
    d=json.load(open("$OUT/%s.json"%n)); print(n, d["value"], d["ms_per_step"])
PY
