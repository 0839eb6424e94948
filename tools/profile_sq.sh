#!/bin/bash
# Two SQ counter passes (8 counters each, counters only with --kernel-trace) of a single-stream bench run:
#   gpurun -- 'bash tools/profile_sq.sh TAG'   ->  gpurun_out/TAG/pmc_sq{,2}/  (summarise with tools/pmc_counters.py)
set -e
TAG=${1:-sq}
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU --output-format csv -d $OUT/pmc_sq -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-profile --contexts 1 > /dev/null 2> $OUT/pmc_sq.err
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD --output-format csv -d $OUT/pmc_sq2 -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-profile --contexts 1 > /dev/null 2> $OUT/pmc_sq2.err
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline --contexts 1 > $OUT/bench_ctx1.json 2> $OUT/bench_ctx1.err
find $OUT -name '*kernel_trace.csv' -delete
find $OUT -name '*agent_info.csv' -delete
python3 $R/tools/pmc_counters.py $OUT/pmc_sq $OUT/pmc_sq2 > $OUT/counters.csv
cat $OUT/counters.csv | cut -c1-400 | head -12
