#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/r03
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
AMD_LOG_LEVEL=4 timeout -k 10 60 $R/tools/micro/d2h_engine hip 2>&1 | grep -i "HSA Copy\|Query copy engine" | sed 's/dst=0x[0-9a-f]*/dst=X/; s/src=0x[0-9a-f]*/src=X/; s/wait_event=0x[0-9a-f]*/w=X/; s/completion_signal=0x[0-9a-f]*/c=X/; s/[0-9]\{6,\} us/N us/; s/tid: 0x[0-9a-f]*/tid/; s/Agent 0x[0-9a-f]*/Agent X/g' | sort | uniq -c | sort -rn | head -12 > $OUT/eng_hip_log.txt; cat $OUT/eng_hip_log.txt
for e in 0 1 2 3 4 5; do
  timeout -k 10 60 $R/tools/micro/d2h_engine hsaeng$e > $OUT/eng_hsaeng$e.txt 2>&1; echo "== engine bit $e"; tail -5 $OUT/eng_hsaeng$e.txt
done
cd $R
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $OUT/pytest_b.log 2>&1; echo "pytest rc=$?"; tail -5 $OUT/pytest_b.log
for v in march tiles march tiles; do
  if [ $v = tiles ]; then export HESS_LIB=$R/tools/_variants/tiles/libhessgpu.so; else unset HESS_LIB; fi
  HESS_DELIVERY=mirror timeout -k 10 300 python bench.py --no-cpu-baseline --no-api-leg --no-configs4 > $OUT/gauss_${v}_$(date +%s).json 2> $OUT/gauss_${v}.err; echo "bench $v rc=$?"
done
unset HESS_LIB
ls $OUT | head -50
