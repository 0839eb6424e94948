#!/bin/bash
# Number of pipelined contexts, on the driver's command (--steps 20 --warmup 5: `value` and the steady-state leg's figure)
# and on the default 200 steps.  tools/r05/r05_ctx_sweep.sh "4 5 6 7 8 9"
for rnd in 1 2; do
for c in $1; do
  python bench.py --steps 20 --warmup 5 --contexts $c --no-cpu-baseline --no-api-leg --no-configs4 --no-host-leg --no-profile 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('contexts $c: driver command value', d['value'], 'steady', d.get('value_steady_state'), 'ratio %.3f' % (d['value']/d['value_steady_state']))"
done
done
