#!/bin/bash
# the SiftGPU class from 1 / 8 host threads (apps/multithread.cpp -mem) under schedule switches of the developer build
R=${GRAFT_REPO_ROOT:-$PWD}
python3 - <<PY
import sys; sys.path.insert(0, "$R/tests"); sys.path.insert(0, "$R")
import fixtures
img = fixtures.synthetic_blobs(1920, 1080, 0)
open("/tmp/bench.pgm", "wb").write(b"P5\n1920 1080\n255\n" + img.tobytes())
PY
export LD_LIBRARY_PATH=$R/hessgpu_amd/dev
for rnd in 1 2; do
for v in "-" "HESS_CHAIN_FROM=99" "HESS_DELIVERY=dma" "HESS_CHAIN_FROM=99 HESS_DELIVERY=dma" "HESS_MIRROR_MAX_BATCH=0"; do
  if [ "$v" = "-" ]; then e=""; else e="$v"; fi
  for k in 1 8; do
    r=$(env $e $R/hessgpu_amd/bin/multithread -i /tmp/bench.pgm -mem -n 150 -devices 1 -per-device $k -topk 4096 2>&1 | grep -o "MPIX: [0-9.]*")
    echo "$v threads $k: $r"
  done
done
done
