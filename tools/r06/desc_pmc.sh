#!/bin/bash
# the two SQ counter passes of tools/profile_round.sh alone (1080p bench workload, one stream): tools/r06/desc_pmc.sh TAG
TAG=${1:-r06_descpmc}
R=${GRAFT_REPO_ROOT:-$PWD}; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
A="--steps 2 --warmup 1 --no-cpu-baseline --no-profile --no-steady --no-host-leg --no-api-leg --no-configs4 --no-real-images --no-matcher --contexts 1"
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU --output-format csv -d $OUT/pmc_sq -- python3 $R/bench.py $A > /dev/null 2> $OUT/pmc_sq.err || exit 3
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD --output-format csv -d $OUT/pmc_sq2 -- python3 $R/bench.py $A > /dev/null 2> $OUT/pmc_sq2.err || exit 4
python3 $R/tools/pmc_counters.py $OUT/pmc_sq $OUT/pmc_sq2 > $OUT/counters.csv
find $OUT -name '*kernel_trace.csv' -delete; find $OUT -name '*agent_info.csv' -delete; find $OUT -name '*counter_collection.csv' -delete
grep -i "name\|descriptor\|orientation" $OUT/counters.csv
