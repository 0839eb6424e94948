#!/bin/bash
# HIP's GPU_MAX_HW_QUEUES (default 4: six contexts' streams share three) on the final kernels: pipelined line, 200 steps and the driver's 20, twice
for rnd in 1 2; do for q in ${QUEUES:-4 6 8}; do
for k in 200 20; do
GPU_MAX_HW_QUEUES=$q python bench.py --steps $k --warmup 5 --no-cpu-baseline --no-api-leg --no-host-leg --no-configs4 --no-real-images --no-matcher --no-steady 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('GPU_MAX_HW_QUEUES=$q steps', d['steps'], 'value', d['value'], 'ms/step', d['ms_per_step'])"
done; done; done
