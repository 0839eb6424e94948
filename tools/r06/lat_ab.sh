#!/bin/bash
# single-image latency of library variants, alternating, twice: tools/r06/lat_ab.sh lib lib ...   ("cur" = in-tree)
R=${GRAFT_REPO_ROOT:-$PWD}
for rnd in 1 2; do for v in "$@"; do
  if [ "$v" = "cur" ]; then unset HESS_LIB; else export HESS_LIB=$R/tools/_variants/$v/libhessgpu.so; fi
  timeout -k 10 200 python3 $R/tools/r06/lat_ab.py || exit 3
done; done
