#!/bin/bash
# pipelined contexts per GPU, same call, 200 steps, twice
for rnd in 1 2; do for n in 5 6 7 8 9; do
python bench.py --steps 200 --contexts $n --no-cpu-baseline --no-api-leg --no-host-leg --no-configs4 --no-real-images --no-matcher --no-steady 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('contexts', $n, 'value', d['value'], 'ms/step', d['ms_per_step'])"
done; done
