#!/bin/bash
# counter passes of the matcher at 8192^2 (each its own run; --kernel-trace only beside --pmc):  tools/r06/match_pmc.sh TAG
TAG=${1:-match_pmc}
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/$TAG; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU --output-format csv -d $OUT/pmc1 -- python3 $R/tools/bench_match.py --only 8192 > /dev/null 2> $OUT/pmc1.err
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_I8 SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA --output-format csv -d $OUT/pmc2 -- python3 $R/tools/bench_match.py --only 8192 > /dev/null 2> $OUT/pmc2.err
rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc3 -- python3 $R/tools/bench_match.py --only 8192 > /dev/null 2> $OUT/pmc3.err
python3 $R/tools/pmc_counters.py $OUT/pmc1 $OUT/pmc2 $OUT/pmc3 --kernels match > $OUT/counters.csv
cat $OUT/counters.csv
tail -3 $OUT/pmc2.err
find $OUT -name '*kernel_trace.csv' -delete; find $OUT -name '*agent_info.csv' -delete
