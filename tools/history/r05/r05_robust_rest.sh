OUT=gpurun_out/r05_robust; mkdir -p $OUT
T="tests/test_gpu_parity.py tests/test_reference_inputs_gpu.py tests/test_keypoint_list_gpu.py tests/test_descriptor_order.py tests/test_shared_results.py"
for v in "HESS_DELIVERY=mirror" "HESS_DELIVERY=blit"; do
  n=$(echo $v | tr '=' '_')
  env $v timeout -k 10 300 python -m pytest $T -m gpu -x -q --durations=5 > $OUT/$n.log 2>&1; echo "$v: $(tail -1 $OUT/$n.log)"
done
timeout -k 10 200 python tools/soak.py 300 host > $OUT/soak_host.log 2>&1; echo "soak host: $(tail -1 $OUT/soak_host.log)"
