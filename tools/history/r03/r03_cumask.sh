#!/bin/bash
# EXPERIMENT: CU masks for the context's stream (HESS_MAIN_CUS) and a separate descriptor stream (HESS_DESC_CUS).
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/r03cu
mkdir -p $OUT; cd $R
run() {  # name, env...
  local name=$1; shift
  env "$@" timeout -k 10 200 python bench.py --steps 100 --no-cpu-baseline --no-api-leg --no-configs4 --no-host-leg > $OUT/$name.json 2> $OUT/$name.err || { echo "$name failed"; tail -3 $OUT/$name.err; return 1; }
  python - $name $OUT/$name.json <<'PY'
import json,sys
d=json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
print(sys.argv[1], d["value"], "ms/step", d["ms_per_step"], {k:round(v,3) for k,v in d.get("kernel_ms_per_step",{}).items()}, "parity", d.get("parity_checked"))
PY
}
run base A=1 &&
run main192 HESS_MAIN_CUS=192 &&
run main128 HESS_MAIN_CUS=128 &&
run desc256 HESS_DESC_CUS=256 &&
run m192_d256 HESS_MAIN_CUS=192 HESS_DESC_CUS=256 &&
run m160_d256 HESS_MAIN_CUS=160 HESS_DESC_CUS=256 &&
run m128_d128 HESS_MAIN_CUS=128 HESS_DESC_CUS=128 &&
run m112_d144 HESS_MAIN_CUS=112 HESS_DESC_CUS=144 &&
run m192_d160 HESS_MAIN_CUS=192 HESS_DESC_CUS=160 &&
run base2 A=1
