#!/bin/bash
# The plugin-surface legs of bench.py (RunSIFT + GetFeatureVector, 1 thread / 8 instances) under environment settings
for rnd in 1 2; do
for v in "$@"; do
  if [ "$v" = "-" ]; then envs=""; else envs="$v"; fi
  env $envs timeout -k 10 300 python bench.py --steps 20 --no-cpu-baseline --no-configs4 --no-host-leg --no-steady --no-profile > gpurun_out/r05_apiq.json 2> gpurun_out/r05_apiq.err || { echo "failed $v"; tail -3 gpurun_out/r05_apiq.err; exit 9; }
  python - "$v" <<'PY'
import json,sys
d=json.loads(open('gpurun_out/r05_apiq.json').read().strip().splitlines()[-1])
print(sys.argv[1], 'api 1 thread', d['value_siftgpu_api_1thread'], ' 8 threads', d['value_siftgpu_api_threads'], ' latency', d.get('latency_ms_single_image'))
PY
done
done
