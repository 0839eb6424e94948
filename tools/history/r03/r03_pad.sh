#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/r03
mkdir -p $OUT; cd $R
for rep in 1 2; do
for pad in default 0 8192 14336; do
  if [ $pad = default ]; then unset HESS_DESC_LDS_PAD; else export HESS_DESC_LDS_PAD=$pad; fi
  timeout -k 10 300 python bench.py --no-cpu-baseline --no-api-leg --no-configs4 --no-host-leg > $OUT/pad_${pad}_$rep.json 2> $OUT/pad.err
  python - <<PY
import json
d=json.load(open("$OUT/pad_${pad}_$rep.json")); print("pad $pad:", d["value"], d["kernel_ms_per_step"]["descriptor"])
PY
done
done
unset HESS_DESC_LDS_PAD
HESS_BENCH_FORCE_DIST=1 RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 MASTER_ADDR=127.0.0.1 MASTER_PORT=29511 timeout -k 10 300 python bench.py --no-api-leg --no-configs4 --no-host-leg > $OUT/forcedist_host.json 2> $OUT/forcedist.err; python -c "
import json; d=json.load(open('$OUT/forcedist_host.json')); print('force dist (host landing):', d['value'], d.get('parity_checked'), d['config']['sharding'])"
HESS_BENCH_FORCE_DIST=1 RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 MASTER_ADDR=127.0.0.1 MASTER_PORT=29512 timeout -k 10 300 python bench.py --gather-dest hbm --no-api-leg --no-configs4 --no-host-leg --no-cpu-baseline > $OUT/forcedist_hbm.json 2>> $OUT/forcedist.err; python -c "
import json; d=json.load(open('$OUT/forcedist_hbm.json')); print('force dist (hbm):', d['value'])"
timeout -k 10 300 python bench.py --no-api-leg --no-configs4 --no-host-leg --no-cpu-baseline > $OUT/nodist.json 2>> $OUT/forcedist.err; python -c "
import json; d=json.load(open('$OUT/nodist.json')); print('no dist:', d['value'])"
tail -3 $OUT/forcedist.err
