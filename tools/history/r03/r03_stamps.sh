#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/r03
mkdir -p $OUT
cd $R
for v in stamps_tiles stamps_march; do
  for o in 1 7; do
    HESS_LIB=$R/tools/_variants/$v/libhessgpu.so timeout -k 10 200 python tools/gauss_stamps.py --octaves $o > $OUT/${v}_o$o.txt 2>&1; cat $OUT/${v}_o$o.txt | tail -9
  done
done
