#!/bin/bash
# A/B bench of developer variants (tools/_variants/<name>/libhessgpu.so, built by `python -m hessgpu_amd.build
# --variant NAME flags...`) on the GPU box:  tools/ab_bench.sh TAG "pytest args or -" variant...   ("cur" = in-tree lib)
# Runs the given GPU tests first with the in-tree library; benches only if pytest did not crash (rc 0 or 1).
TAG=$1; TESTS=$2; shift 2
mkdir -p gpurun_out
if [ "$TESTS" != "-" ]; then
  timeout -k 10 600 python -m pytest $TESTS -m gpu -x -q > gpurun_out/${TAG}_tests.log 2>&1
  rc=$?; echo "pytest rc=$rc" >> gpurun_out/${TAG}_tests.log; tail -4 gpurun_out/${TAG}_tests.log
  if [ $rc -gt 1 ]; then echo "pytest crashed: no bench"; exit $rc; fi
fi
for v in "$@"; do
  if [ "$v" = "cur" ]; then unset HESS_LIB; else export HESS_LIB=$PWD/tools/_variants/$v/libhessgpu.so; fi
  timeout -k 10 300 python bench.py --steps 20 --no-cpu-baseline ${BENCH_ARGS} > gpurun_out/${TAG}_$v.json 2> gpurun_out/${TAG}_$v.err || { echo "bench $v failed"; tail -3 gpurun_out/${TAG}_$v.err; exit 9; }
  python - "$v" gpurun_out/${TAG}_$v.json <<'PY'
import json,sys
d=json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
print(sys.argv[1], d["value"], "ms/step", d["ms_per_step"], {k:round(v,3) for k,v in d.get("kernel_ms_per_step",{}).items()})
PY
done
