#!/bin/bash
# FETCH_SIZE per launch of the kernels matching $1 for library builds $2.. (same call): tools/r04/r04_fetch.sh extrema cur variant
R=${GRAFT_REPO_ROOT:-$PWD}; K=$1; shift; cd /tmp && export TMPDIR=/tmp
for v in "$@"; do
  if [ "$v" = "cur" ]; then unset HESS_LIB; else export HESS_LIB=$R/tools/_variants/$v/libhessgpu.so; fi
  rm -rf /tmp/fetch_$v
  rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d /tmp/fetch_$v -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-profile --no-steady --no-host-leg --no-api-leg --no-configs4 --contexts 1 > /dev/null 2>&1
  echo "== $v"; python3 $R/tools/pmc_counters.py /tmp/fetch_$v --kernels $K | cut -c1-120
done
