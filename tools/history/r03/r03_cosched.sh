#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/r03
mkdir -p $OUT; cd $R
A="--steps 300 --warmup 10 --no-cpu-baseline --no-api-leg --no-configs4 --no-profile --no-host-leg"
for dp in 0 19456 23552 36000; do
for gp in 0 8192 13000; do
  HESS_DESC_LDS_PAD=$dp HESS_GAUSS_LDS_PAD=$gp python bench.py $A > $OUT/cs_${dp}_${gp}.json 2>/dev/null
  python -c "
import json; d=json.load(open('$OUT/cs_${dp}_${gp}.json')); print('desc pad $dp gauss pad $gp:', d['value'])"
done
done
