#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/r03
mkdir -p $OUT; cd $R
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $OUT/pytest_d.log 2>&1; echo "pytest rc=$?"; tail -4 $OUT/pytest_d.log
for rep in 1 2; do
for v in cur prev; do
  if [ "$v" = "cur" ]; then unset HESS_LIB; else export HESS_LIB=$R/tools/_variants/$v/libhessgpu.so; fi
  timeout -k 10 300 python bench.py --no-cpu-baseline --no-api-leg --no-configs4 --no-host-leg > $OUT/ext_${v}_$rep.json 2> $OUT/ext.err
  python - <<PY
import json
d=json.load(open("$OUT/ext_${v}_$rep.json")); print("$v", d["value"], d["kernel_ms_per_step"])
PY
done
done
cd /tmp && export TMPDIR=/tmp
for v in cur prev; do
  if [ "$v" = "cur" ]; then unset HESS_LIB; else export HESS_LIB=$R/tools/_variants/$v/libhessgpu.so; fi
  for c in FETCH_SIZE WRITE_SIZE; do
    timeout -k 10 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $OUT/pmc_${v}_$c -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-profile --no-host-leg --no-api-leg --no-configs4 --contexts 1 > /dev/null 2> $OUT/pmc_$v.err
  done
  python3 $R/tools/pmc_traffic.py $OUT/pmc_${v}_FETCH_SIZE $OUT/pmc_${v}_WRITE_SIZE > $OUT/traffic_$v.json
  python3 - <<PY
import json
d=json.load(open("$OUT/traffic_$v.json"))
for k in d["per_kernel"]:
    if "extrema" in k["kernel"]: print("$v", k["kernel"], round(k["hbm_bytes_per_launch_corrected"]/1e6,1), "MB/launch")
PY
  find $OUT/pmc_${v}_* -name '*kernel_trace.csv' -delete
done
