#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/r03api
mkdir -p $OUT; cd $R
timeout -k 10 600 python -m pytest tests/test_siftgpu_engine_gpu.py tests/test_siftgpu_api_gpu.py tests/test_keypoint_list_gpu.py tests/test_cli_gpu.py -x -q 2>&1 | tail -4
python - <<'PY'
import sys
sys.path.insert(0, "tests")
import fixtures
img = fixtures.synthetic_blobs(1920, 1080, 0)
with open("gpurun_out/r03api/bench.pgm", "wb") as f:
    f.write(b"P5\n1920 1080\n255\n"); f.write(img.tobytes())
PY
for rep in 1 2; do
for k in 1 2 4 8 12 16; do
  for b in 1 0; do
    r=$(HESS_API_BATCH=$b timeout -k 10 120 hessgpu_amd/bin/multithread -i $OUT/bench.pgm -mem -n 200 -devices 1 -per-device $k -topk 4096 2>&1 | grep -o "MPIX: [0-9.]*\|differ\|FAIL.*")
    echo "threads=$k engine=$b $r"
  done
done
done
