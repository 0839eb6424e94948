#!/bin/bash
# the driver's command N times with the completion stamps; prints every run, and the gaps of the slow ones: tools/r06/driver_outliers.sh [N]
for i in $(seq 1 ${1:-30}); do
HESS_BENCH_STAMPS=1 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-api-leg --no-host-leg --no-configs4 --no-real-images --no-matcher --no-steady 2> /tmp/out_$i.err | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print($i, d['value'], d['ms_per_step'])" | tee /tmp/out_$i.txt
v=$(cut -d' ' -f2 /tmp/out_$i.txt | cut -d. -f1)
if [ "$v" -lt 18500 ]; then grep "completion gaps\|submitting thread" /tmp/out_$i.err | cut -c1-260; fi
done
