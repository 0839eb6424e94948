#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/r03
mkdir -p $OUT
cd $R
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -k "delivery or grows or refused or pageable" > $OUT/pytest_c.log 2>&1; echo "pytest rc=$?"; tail -4 $OUT/pytest_c.log
for rep in 1 2; do
for m in mirror dma dmahip; do
  if [ $m = dmahip ]; then export HESS_COPIER=hip; d=dma; else unset HESS_COPIER; d=$m; fi
  HESS_DELIVERY=$d timeout -k 10 300 python bench.py --no-cpu-baseline --no-api-leg --no-configs4 > $OUT/sdma_${m}_$rep.json 2> $OUT/sdma_$m.err
  python - <<PY
import json
d=json.load(open("$OUT/sdma_${m}_$rep.json")); print("$m", d["value"], d["value_host_to_host"], d["latency_ms_single_image"], d["kernel_ms_per_step"]["descriptor"])
PY
done
done
unset HESS_COPIER
cd /tmp && export TMPDIR=/tmp
HESS_DELIVERY=dma timeout -k 10 300 rocprofv3 --kernel-trace --memory-copy-trace --stats --output-format csv -d $OUT/stats_sdma -- python3 $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-host-leg --no-api-leg --no-configs4 > $OUT/prof_sdma.json 2> $OUT/prof_sdma.err; echo "rocprof rc=$?"
grep -c copyBuffer $OUT/stats_sdma/*/*kernel_trace.csv
cat $OUT/stats_sdma/*/*memory_copy_stats.csv 2>/dev/null | head
find $OUT/stats_sdma -name '*_trace.csv' -size +5M -delete
