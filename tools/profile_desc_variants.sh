#!/bin/bash
# SQ + cache counters of the descriptor kernel for developer variants (tools/_variants/<name>):
#   gpurun -- 'bash tools/profile_desc_variants.sh TAG variant...'   ("cur" = in-tree library)
set -e
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-$PWD}
cd /tmp && export TMPDIR=/tmp
for v in "$@"; do
  if [ "$v" = "cur" ]; then unset HESS_LIB; else export HESS_LIB=$R/tools/_variants/$v/libhessgpu.so; fi
  OUT=$R/gpurun_out/$TAG/$v
  mkdir -p $OUT
  rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU --output-format csv -d $OUT/pmc_sq -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-profile --no-host-leg --contexts 1 > /dev/null 2> $OUT/pmc_sq.err
  rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM --output-format csv -d $OUT/pmc_sq2 -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-profile --no-host-leg --contexts 1 > /dev/null 2> $OUT/pmc_sq2.err
  rocprofv3 --kernel-trace --pmc SQ_WAVES TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum --output-format csv -d $OUT/pmc_tcp -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-profile --no-host-leg --contexts 1 > /dev/null 2> $OUT/pmc_tcp.err || echo "tcp pass failed for $v"
  python3 $R/tools/pmc_counters.py $OUT/pmc_sq $OUT/pmc_sq2 $OUT/pmc_tcp --kernels descriptor > $OUT/counters.csv
  find $OUT -name '*kernel_trace.csv' -delete; find $OUT -name '*agent_info.csv' -delete
  echo "== $v"; cat $OUT/counters.csv
done
