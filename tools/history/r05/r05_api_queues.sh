#!/bin/bash
# The plugin-surface legs of bench.py (SiftGPU::RunSIFT + GetFeatureVector, 1 thread / 8 instances) by GPU_MAX_HW_QUEUES
for rnd in 1 2; do
for q in 4 8 12; do
  GPU_MAX_HW_QUEUES=$q timeout -k 10 300 python bench.py --steps 50 --no-cpu-baseline --no-configs4 --no-host-leg --no-steady > gpurun_out/r05_apiq.json 2> gpurun_out/r05_apiq.err || { echo "failed $q"; tail -3 gpurun_out/r05_apiq.err; exit 9; }
  python - $q <<'PY'
import json,sys
d=json.loads(open('gpurun_out/r05_apiq.json').read().strip().splitlines()[-1])
print('queues',sys.argv[1],'value',d['value'],{k:v for k,v in d.items() if 'api' in k or 'latency' in k})
PY
done
done
