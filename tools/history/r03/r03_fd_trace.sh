#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/r03fd
rm -rf $OUT; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $OUT/t -- python3 $R/tools/r03/r03_fill_drain.py 20 > $OUT/run.txt 2> $OUT/run.err
tail -12 $OUT/run.txt
python3 $R/tools/r03/r03_fd_trace.py $OUT/t | tail -30
