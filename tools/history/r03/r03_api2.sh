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
for rep in 1 2 3; do
for k in 1 6 8 10; do
  r=$(timeout -k 10 120 hessgpu_amd/bin/multithread -i $OUT/bench.pgm -mem -n 150 -devices 1 -per-device $k -topk 4096 2>/dev/null | grep -o "MPIX: [0-9.]*")
  echo "threads=$k $r"
done
done
