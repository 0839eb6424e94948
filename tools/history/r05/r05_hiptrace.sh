# Host API calls + kernels + copies of the pipelined bench on one time axis (gpurun_out/r05_ht/*.csv)
R=${GRAFT_REPO_ROOT:-$PWD}; OUT=$R/gpurun_out/r05_ht; mkdir -p $OUT; cd /tmp && export TMPDIR=/tmp
rocprofv3 --hip-runtime-trace --kernel-trace --memory-copy-trace --output-format csv -d /tmp/ht6 -- python3 $R/bench.py --steps 40 --warmup 6 --no-cpu-baseline --no-host-leg --no-api-leg --no-configs4 --no-profile --no-steady ${BENCH_ARGS} > $OUT/bench.json 2> $OUT/bench.err
for f in $(find /tmp/ht6 -name '*.csv'); do echo $f $(wc -l < $f); done
cp $(find /tmp/ht6 -name '*kernel_trace.csv' | head -1) $OUT/kernel_trace.csv
cp $(find /tmp/ht6 -name '*memory_copy_trace.csv' | head -1) $OUT/memory_copy_trace.csv
# the API trace is large: keep the second half
f=$(find /tmp/ht6 -name '*hip_api_trace.csv' | head -1); n=$(wc -l < $f); (head -1 $f; tail -n $((n/2)) $f) | gzip > $OUT/hip_api_trace.csv.gz
ls -la $OUT
