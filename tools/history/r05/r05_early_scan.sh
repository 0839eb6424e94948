mkdir -p gpurun_out/r05_l
HESS_EARLY_SCAN=1 python -m pytest tests/test_gpu_parity.py tests/test_reference_inputs_gpu.py -x -q -m gpu > gpurun_out/r05_l/pytest_early.txt 2>&1; tail -3 gpurun_out/r05_l/pytest_early.txt
bash tools/r05/r05_ab_env.sh r05_l/early "-" "HESS_EARLY_SCAN=1" 2>&1 | tee gpurun_out/r05_l/ab_early.txt
export HESS_LIB=$PWD/tools/_variants/plain/libhessgpu.so
bash tools/r05/r05_ab_env.sh r05_l/early_plain "-" "HESS_EARLY_SCAN=1" 2>&1 | tee gpurun_out/r05_l/ab_early_plain.txt
