#!/bin/bash
# Profiling passes of one round, run on the GPU box:  gpurun -- 'bash tools/profile_round.sh r01_c'
# Writes under gpurun_out/<tag>/ ; copy the summaries into profiles/ afterwards (see profiles/README.md).
set -e
TAG=${1:-prof}
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
# 1. kernel statistics, single stream (averages comparable with bench.py's hipEvent roofline leg)
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_ctx1 -- python3 $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline --contexts 1 > $OUT/bench_ctx1.json 2> $OUT/bench_ctx1.err
# 2. kernel statistics, default pipelined contexts
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_default -- python3 $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline > $OUT/bench_default.json 2> $OUT/bench_default.err
# 3./4. HBM traffic counters, one pass each
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-profile --contexts 1 > /dev/null 2> $OUT/pmc_fetch.err
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-profile --contexts 1 > /dev/null 2> $OUT/pmc_write.err
# 5. wavefront occupancy of time (wait / issue-stall / active), one pass of 8 SQ counters
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU --output-format csv -d $OUT/pmc_sq -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-profile --contexts 1 > /dev/null 2> $OUT/pmc_sq.err
# keep only the small summaries (the traces are large)
find $OUT -name '*kernel_trace.csv' -delete
find $OUT -name '*agent_info.csv' -delete
ls -R $OUT | head -40
