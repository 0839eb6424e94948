#!/bin/bash
# the driver's command with its steady-state leg, N times: value, value_steady_state, ratio
for i in $(seq 1 ${1:-6}); do
python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-api-leg --no-host-leg --no-configs4 --no-real-images --no-matcher 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print(d['value'], d['value_steady_state'], round(d['value']/d['value_steady_state'],3))"
done
