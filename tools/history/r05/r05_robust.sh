#!/bin/bash
# Robustness passes on the final code of round 5 (GPU box): the GPU parity suite under every delivery form and A/B switch
# (incl. this round's: the top level stored + its own det-H launch, the early scan, small XCD blocks, the float descriptor
# orders as defaults are covered by tests/test_descriptor_order.py), the randomised parity sweep, the determinism soak.
OUT=gpurun_out/r05_robust; mkdir -p $OUT
T="tests/test_gpu_parity.py tests/test_reference_inputs_gpu.py tests/test_keypoint_list_gpu.py tests/test_descriptor_order.py tests/test_shared_results.py"
# PART=1: the first eight switches; PART=2: the others + fuzz + soak (a slow box does not finish all of it inside one 1200 s call)
V1="HESS_DELIVERY=mirror HESS_DELIVERY=blit HESS_DELIVERY=dma HESS_CHAIN_FROM=2 HESS_CHAIN_FROM=99 HESS_NO_PAIR=1 HESS_SCATTER_SCAN=1 HESS_SCATTER_SCAN=0"
V2="HESS_COPIER=hip HESS_DESC_XCD=0 HESS_DESC_XCD=1 HESS_DESC_XCD=3 HESS_NO_TOP_FUSION=1 HESS_NO_FIRST_FUSION=1 HESS_EARLY_SCAN=1 HESS_NO_PRIME_BATCH=1 HESS_MIRROR_MAX_MB=0"
case "${PART:-0}" in 1) VS="$V1";; 2) VS="$V2";; *) VS="$V1 $V2";; esac
for v in $VS; do
  n=$(echo $v | tr '=' '_')
  env $v timeout -k 10 400 python -m pytest $T -m gpu -x -q > $OUT/$n.log 2>&1; echo "$v: $(tail -1 $OUT/$n.log)"
done
[ "${PART:-0}" = "1" ] && exit 0
timeout -k 10 1000 python tools/fuzz_parity.py ${FUZZ:-300} ${SEED:-5} > $OUT/fuzz.log 2>&1; echo "fuzz: $(tail -1 $OUT/fuzz.log)"
timeout -k 10 300 python tools/soak.py 300 > $OUT/soak.log 2>&1; echo "soak: $(tail -1 $OUT/soak.log)"
timeout -k 10 300 python tools/soak.py 300 host > $OUT/soak_host.log 2>&1; echo "soak host: $(tail -1 $OUT/soak_host.log)"
