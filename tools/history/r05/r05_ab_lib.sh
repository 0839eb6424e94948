#!/bin/bash
# Same-call A/B of library builds (tools/_variants/<name>/libhessgpu.so, "cur" = the in-tree one):
#   tools/r05/r05_ab_lib.sh TAG lib lib ...    bench.py --steps 200 without the side legs, twice per library, alternating
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-$PWD}
mkdir -p gpurun_out
for rnd in 1 2; do
for v in "$@"; do
  if [ "$v" = "cur" ]; then unset HESS_LIB; else export HESS_LIB=$R/tools/_variants/$v/libhessgpu.so; fi
  timeout -k 10 300 python bench.py --steps 200 --no-cpu-baseline --no-api-leg --no-configs4 --no-host-leg ${BENCH_ARGS} > gpurun_out/${TAG}_$v.json 2> gpurun_out/${TAG}_$v.err || { echo "bench $v failed"; tail -3 gpurun_out/${TAG}_$v.err; exit 9; }
  python - "$v" gpurun_out/${TAG}_$v.json <<'PY'
import json,sys
d=json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
print(sys.argv[1], d["value"], "ms/step", d["ms_per_step"], {k:round(v,3) for k,v in d.get("kernel_ms_per_step",{}).items()}, "parity", d.get("parity_checked"))
PY
done
done
