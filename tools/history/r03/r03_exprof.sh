#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/r03exprof
mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
for v in depth1 d2_uncond qend_uncond; do
  export HESS_LIB=$R/tools/_variants/$v/libhessgpu.so
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/$v -- python3 $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-host-leg --no-api-leg --no-configs4 --contexts 1 > $OUT/$v.json 2> $OUT/$v.err
  f=$(ls $OUT/$v/*/*_kernel_stats.csv | head -1)
  echo "== $v"; grep -E "extrema_stream|extrema_scatter|row_scan" $f | sed 's/.*(anonymous namespace):://' | awk -F'",' '{print substr($1,1,28), $2}' | cut -c1-100
done
