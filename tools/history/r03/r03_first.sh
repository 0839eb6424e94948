#!/bin/bash
# Round 3, first GPU call: tests, VALU peak table, delivery A/B (same box, same call), kernel trace of the default bench.
set -o pipefail
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/r03
mkdir -p $OUT
cd $R
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $OUT/pytest_a.log 2>&1; echo "pytest rc=$?" | tee -a $OUT/pytest_a.log
tail -5 $OUT/pytest_a.log
timeout -k 10 200 tools/micro/valu_peak > $OUT/valu_peak.txt 2>&1; echo "valu_peak rc=$?"
for mode in dma mirror blit dma mirror; do
  HESS_DELIVERY=$mode timeout -k 10 300 python bench.py --no-cpu-baseline --no-api-leg --no-configs4 > $OUT/ab_${mode}_$(date +%s).json 2> $OUT/ab_${mode}.err; echo "bench $mode rc=$?"
done
timeout -k 10 600 python bench.py > $OUT/bench_default_a.json 2> $OUT/bench_default_a.err; echo "bench default rc=$?"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_default -- python3 $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-host-leg --no-api-leg --no-configs4 > $OUT/bench_ctx3_prof.json 2> $OUT/bench_ctx3_prof.err; echo "rocprof rc=$?"
find $OUT -name '*kernel_trace.csv' -size +20M -delete
ls $OUT
