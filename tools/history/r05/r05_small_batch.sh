#!/bin/bash
# Pipelined rate of small batches (device-resident) under the latency-mode switches: tools/r05/r05_small_batch.sh BATCH CONTEXTS "ENV..." ...
B=$1; C=$2; shift 2
for v in "$@"; do
  if [ "$v" = "-" ]; then envs=""; else envs="$v"; fi
  env $envs python bench.py --batch $B --contexts $C --steps 400 --no-cpu-baseline --no-api-leg --no-configs4 --no-host-leg --no-steady --no-profile 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('batch $B contexts $C [$v]:', d['value'], d['ms_per_step'])"
done
