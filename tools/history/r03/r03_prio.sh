#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/r03prio
mkdir -p $OUT; cd $R
run() {
  local name=$1; shift
  env "$@" timeout -k 10 200 python bench.py --steps 120 --no-cpu-baseline --no-api-leg --no-configs4 --no-host-leg --no-profile > $OUT/$name.json 2> $OUT/$name.err || { echo "$name failed"; tail -3 $OUT/$name.err; return 1; }
  python - $name $OUT/$name.json <<'PY'
import json,sys
d=json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
print(sys.argv[1], d["value"], "ms/step", d["ms_per_step"])
PY
}
HESS_AB_PRIO_PRINT=1 run base A=1 && grep -h "priority range" $OUT/*.err | head -1
run p_hi_lo HESS_AB_PRIO_PRINT=1 HESS_AB_PRIO=-1,0 && grep -h "priority range" $OUT/p_hi_lo.err | head -1
run p_all_hi HESS_AB_PRIO=-1 && run p_3lvl HESS_AB_PRIO=-1,0,1 && run p_hi_lo_lo HESS_AB_PRIO=-1,0,0 && run p_desc HESS_AB_PRIO=-1,-1,-1,0,0,0 && run base2 A=1 && run p_hi_lo2 HESS_AB_PRIO=-1,0
