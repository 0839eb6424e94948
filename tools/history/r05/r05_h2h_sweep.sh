#!/bin/bash
# Host-to-host rate (pinned pixels in, results out) against the number of pipelined contexts: tools/r05/r05_h2h_sweep.sh "6 8 10 12"
for rnd in 1 2; do
for c in $1; do
  python bench.py --steps 200 --contexts $c --no-cpu-baseline --no-api-leg --no-configs4 --no-profile 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('contexts $c: device-resident', d['value'], ' host-to-host', d['value_host_to_host'], ' single image ms', d['latency_ms_single_image'])"
done
done
