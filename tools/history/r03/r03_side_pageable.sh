#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/r03api
mkdir -p $OUT; cd $R
python - <<'PY'
import sys
sys.path.insert(0, "tests")
import fixtures
img = fixtures.synthetic_blobs(1920, 1080, 0)
with open("gpurun_out/r03api/bench.pgm", "wb") as f:
    f.write(b"P5\n1920 1080\n255\n"); f.write(img.tobytes())
PY
for rep in 1 2; do
for s in 0 1; do
  if [ $s = 1 ]; then export HESS_AB_SIDE_PAGEABLE=1; else unset HESS_AB_SIDE_PAGEABLE; fi
  for k in 1 4 8; do
    r=$(timeout -k 10 120 hessgpu_amd/bin/multithread -i $OUT/bench.pgm -mem -n 200 -devices 1 -per-device $k -topk 4096 2>/dev/null | grep -o "MPIX: [0-9.]*")
    echo "side=$s threads=$k $r"
  done
  timeout -k 10 200 python tools/bench_host_path.py 2>/dev/null | python -c "
import json,sys
h=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('side=$s', {k:(v['device_resident']['ms_per_batch'], v['host_pinned']['ms_per_batch'], v['host_pageable']['ms_per_batch']) for k,v in h.items() if k.startswith('batch')})"
done
done
