#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/r03h2d
mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --memory-copy-trace --stats --output-format csv -d $OUT/t -- python3 $R/bench.py --steps 30 --warmup 4 --no-cpu-baseline --no-api-leg --no-configs4 --no-profile > $OUT/b.json 2> $OUT/b.err
f=$(ls $OUT/t/*/*_kernel_stats.csv | head -1); m=$(ls $OUT/t/*/*_memory_copy_stats.csv | head -1)
grep -i "copyBuffer\|fill" $f | cut -c1-160; cat $m | cut -c1-160
python3 - <<PY
import json
d=json.loads(open("$OUT/b.json").read().strip().splitlines()[-1]); print(d["value"], d["value_host_to_host"], d["latency_ms_single_image"])
PY
