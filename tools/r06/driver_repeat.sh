#!/bin/bash
# the driver's command N times in one call (spread of the 20-step value): tools/r06/driver_repeat.sh [N]
for i in $(seq 1 ${1:-12}); do
python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-api-leg --no-host-leg --no-configs4 --no-real-images --no-matcher --no-steady 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'])"
done
