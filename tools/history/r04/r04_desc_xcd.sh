#!/bin/bash
# Descriptor launch: XCD-aware feature blocks (HESS_DESC_XCD=block size, 0 = plain order).  Pipelined rate, single-stream
# kernel time and the kernel's FETCH_SIZE per launch, same call.
R=${GRAFT_REPO_ROOT:-$PWD}; OUT=$R/gpurun_out/r04_desc_xcd; mkdir -p $OUT
cd $R
bash tools/r04/r04_ab_env.sh r04dx HESS_DESC_XCD=0 HESS_DESC_XCD=64 HESS_DESC_XCD=256
cd /tmp && export TMPDIR=/tmp
for v in 0 64 256; do
  HESS_DESC_XCD=$v rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/fetch_$v -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-profile --no-steady --no-host-leg --no-api-leg --no-configs4 --contexts 1 > /dev/null 2> $OUT/fetch_$v.err
  echo "HESS_DESC_XCD=$v"; python3 $R/tools/pmc_counters.py $OUT/fetch_$v --kernels descriptor | cut -c1-200
  find $OUT/fetch_$v -name '*kernel_trace.csv' -delete; find $OUT/fetch_$v -name '*counter_collection.csv' -delete
done
