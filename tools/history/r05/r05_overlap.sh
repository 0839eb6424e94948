# Kernel trace of the pipelined bench -> tools/trace_overlap.py (the trace itself is kept: gpurun_out/r05_ov/kernel_trace.csv)
R=${GRAFT_REPO_ROOT:-$PWD}; OUT=$R/gpurun_out/r05_ov; mkdir -p $OUT; cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d /tmp/ov6 -- python3 $R/bench.py --steps 60 --warmup 6 --no-cpu-baseline --no-host-leg --no-api-leg --no-configs4 --no-profile --no-steady ${BENCH_ARGS} > /dev/null 2>&1
cp $(find /tmp/ov6 -name '*kernel_trace.csv' | head -1) $OUT/kernel_trace.csv
python3 $R/tools/trace_overlap.py /tmp/ov6 > $OUT/overlap_contexts6.txt 2>&1; cat $OUT/overlap_contexts6.txt | grep -v "^gauss\|^desc\|^extr\|^orient\|^topk\|^row\|^feat\|^__amd"
