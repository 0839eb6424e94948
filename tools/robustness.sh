#!/bin/bash
# Robustness matrix (GPU box): the GPU parity tests under every delivery form and every developer switch, WITHOUT -x (a red
# line does not hide the tests behind it), every run's tail appended UNEDITED to gpurun_out/<tag>/robust_log.txt -- reruns
# are appended, never substituted.  The switches exist in the developer build only (hessgpu_amd/dev/libhessgpu.so,
# csrc/hess_ctx.h): HESS_TEST_DEV_BUILD=1 makes the suite's context factory use it.
#   tools/robustness.sh TAG [PART]      PART=1: delivery forms + schedule switches, PART=2: the rest + fuzz + soak, default: all
TAG=${1:-robust}; PART=${2:-0}
OUT=gpurun_out/$TAG; mkdir -p $OUT
LOG=$OUT/robust_log.txt
T="tests/test_gpu_parity.py tests/test_reference_inputs_gpu.py tests/test_keypoint_list_gpu.py tests/test_descriptor_order.py tests/test_shared_results.py"
V1="HESS_DELIVERY=mirror HESS_DELIVERY=blit HESS_DELIVERY=dma HESS_CHAIN_FROM=2 HESS_CHAIN_FROM=99 HESS_NO_PAIR=1 HESS_STREAM_ROWS=12"
V2="HESS_PX_BAND=64 HESS_COPIER=hip HESS_DESC_XCD=0 HESS_DESC_XCD=3 HESS_NO_TOP_FUSION=1 HESS_NO_FIRST_FUSION=1 HESS_NO_PRIME_BATCH=1 HESS_MIRROR_MAX_MB=0 HESS_DESC_PARTS=3"
case "$PART" in 1) VS="$V1";; 2) VS="$V2";; *) VS="$V1 $V2";; esac
echo "== $(date -u +%FT%TZ) kernel sources $(python3 -c 'from hessgpu_amd import build; print(build.sources_digest())' 2>/dev/null || echo unknown) (commit $(git rev-parse --short HEAD 2>/dev/null || echo 'n/a on the GPU box')) part $PART" >> $LOG
for v in $VS; do
  n=$(echo $v | tr '=' '_')
  env HESS_TEST_DEV_BUILD=1 $v timeout -k 10 500 python -m pytest $T -m gpu -q > $OUT/$n.log 2>&1
  echo "$v: $(tail -1 $OUT/$n.log)" | tee -a $LOG
  grep -E "^(FAILED|ERROR)" $OUT/$n.log | tee -a $LOG
done
[ "$PART" = "1" ] && exit 0
timeout -k 10 900 python tools/fuzz_parity.py ${FUZZ:-300} ${SEED:-6} > $OUT/fuzz.log 2>&1; echo "fuzz: $(tail -1 $OUT/fuzz.log)" | tee -a $LOG
timeout -k 10 300 python tools/soak.py 300 > $OUT/soak.log 2>&1; echo "soak: $(tail -1 $OUT/soak.log)" | tee -a $LOG
timeout -k 10 300 python tools/soak.py 300 host > $OUT/soak_host.log 2>&1; echo "soak host: $(tail -1 $OUT/soak_host.log)" | tee -a $LOG
