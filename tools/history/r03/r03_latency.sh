#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/r03
mkdir -p $OUT; cd $R
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $OUT/pytest_e.log 2>&1; echo "pytest rc=$?"; tail -3 $OUT/pytest_e.log
python - <<'PY'
import sys; sys.path.insert(0,'tests')
import fixtures, numpy as np
img=fixtures.synthetic_blobs(1920,1080,0)
open('/tmp/bench.pgm','wb').write(b"P5\n1920 1080\n255\n"+img.tobytes())
PY
for k in 1 2 4 6 8 12; do
  timeout -k 10 120 hessgpu_amd/bin/multithread -i /tmp/bench.pgm -mem -n 150 -devices 1 -per-device $k -topk 4096 | grep MPIX
done
for v in cur prev; do
  if [ "$v" = "cur" ]; then unset HESS_LIB; else export HESS_LIB=$R/tools/_variants/$v/libhessgpu.so; fi
  timeout -k 10 300 python tools/bench_host_path.py > $OUT/hostpath_$v.json 2> $OUT/hostpath.err; cat $OUT/hostpath_$v.json
done
