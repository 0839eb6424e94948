#!/bin/bash
# Robustness passes on the final code of round 4 (GPU box): the GPU suite under every delivery form and with the level
# chain forced on batches / off, the randomised parity sweep, and the determinism soak.  Logs under gpurun_out/r04_robust/.
OUT=gpurun_out/r04_robust; mkdir -p $OUT
T="tests/test_gpu_parity.py tests/test_reference_inputs_gpu.py tests/test_keypoint_list_gpu.py tests/test_descriptor_order.py tests/test_shared_results.py"
for v in "HESS_DELIVERY=mirror" "HESS_DELIVERY=blit" "HESS_DELIVERY=dma" "HESS_CHAIN_FROM=2" "HESS_CHAIN_FROM=99" "HESS_NO_PAIR=1" "HESS_SCATTER_SCAN=1" "HESS_SCATTER_SCAN=0" "HESS_COPIER=hip" "HESS_DESC_XCD=0"; do
  n=$(echo $v | tr '=' '_')
  env $v timeout -k 10 400 python -m pytest $T -m gpu -x -q > $OUT/$n.log 2>&1; echo "$v: $(tail -1 $OUT/$n.log)"
done
timeout -k 10 900 python tools/fuzz_parity.py ${FUZZ:-300} 4 > $OUT/fuzz.log 2>&1; echo "fuzz: $(tail -1 $OUT/fuzz.log)"
timeout -k 10 300 python tools/soak.py 300 > $OUT/soak.log 2>&1; echo "soak: $(tail -1 $OUT/soak.log)"
timeout -k 10 300 python tools/soak.py 300 host > $OUT/soak_host.log 2>&1; echo "soak host: $(tail -1 $OUT/soak_host.log)"
