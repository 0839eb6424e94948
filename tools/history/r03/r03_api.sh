#!/bin/bash
# SiftGPU-API thread leg under delivery / queue settings
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/r03api
mkdir -p $OUT; cd $R
python - <<'PY'
import sys, os
sys.path.insert(0, "tests")
import fixtures
img = fixtures.synthetic_blobs(1920, 1080, 0)
with open("gpurun_out/r03api/bench.pgm", "wb") as f:
    f.write(b"P5\n1920 1080\n255\n"); f.write(img.tobytes())
PY
run() {
  local name=$1 k=$2; shift 2
  local r=$(env "$@" timeout -k 10 120 hessgpu_amd/bin/multithread -i $OUT/bench.pgm -mem -n 150 -devices 1 -per-device $k -topk 4096 2>/dev/null | grep -o "MPIX: [0-9.]*")
  echo "$name threads=$k $r"
}
for k in 1 4 8 12; do
  run default $k A=1 && run dma $k HESS_MIRROR_MAX_BATCH=0 && run q8 $k GPU_MAX_HW_QUEUES=8 && run q8dma $k GPU_MAX_HW_QUEUES=8 HESS_MIRROR_MAX_BATCH=0 || exit 1
done
run default 8 A=1
