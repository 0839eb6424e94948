#!/bin/bash
# threads in one process vs the same number of single-thread processes (separate runtimes: no shared host locks)
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
for np in 2 4; do
  for tp in 1 2 4; do
    pids=""
    for i in $(seq 1 $np); do
      timeout -k 10 120 hessgpu_amd/bin/multithread -i $OUT/bench.pgm -mem -n 300 -devices 1 -per-device $tp -topk 4096 > $OUT/p_${np}_${tp}_$i.txt 2>/dev/null &
      pids="$pids $!"
    done
    for p in $pids; do wait $p; done
    echo "processes=$np threads-each=$tp: $(grep -ho 'MPIX: [0-9.]*' $OUT/p_${np}_${tp}_*.txt | awk '{s+=$2} END {print s}') Mpix/s summed"
  done
done
