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
for k in 2 4 8 16; do
  for v in off eb2 eb3 cur eb8; do
    if [ $v = off ]; then export HESS_API_BATCH=0; unset LD_LIBRARY_PATH; elif [ $v = cur ]; then unset HESS_API_BATCH; unset LD_LIBRARY_PATH; else unset HESS_API_BATCH; export LD_LIBRARY_PATH=$R/tools/_variants/$v; fi
    r=$(timeout -k 10 120 hessgpu_amd/bin/multithread -i $OUT/bench.pgm -mem -n 200 -devices 1 -per-device $k -topk 4096 2>&1 | grep -o "MPIX: [0-9.]*\|differ\|FAIL.*")
    echo "threads=$k $v $r"
  done
done
