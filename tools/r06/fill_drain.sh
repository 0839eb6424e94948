#!/bin/bash
# what the timed region of the driver's command (20 steps after 5 warm-up steps) loses against the steady state:
# the same line at 20 / 40 / 80 / 200 steps, twice
for rnd in 1 2; do
for k in 20 40 80 200; do
python bench.py --steps $k --warmup 5 --no-cpu-baseline --no-api-leg --no-host-leg --no-configs4 --no-real-images --no-matcher --no-steady 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('steps', d['steps'], 'ms/step', d['ms_per_step'], 'total ms', round(d['ms_per_step']*d['steps'],2), 'value', d['value'])"
done
done
