#!/bin/bash
# Repeated pipelined runs with chain stamps: is there a slow mode, and what differs in it?
mkdir -p gpurun_out/r05_modes
for i in 1 2 3 4 5 6 7 8 9 10; do
  HESS_CHAIN_STAMPS=1 timeout -k 10 200 python bench.py --steps 200 --no-cpu-baseline --no-api-leg --no-configs4 --no-host-leg --no-steady --no-profile > gpurun_out/r05_modes/run$i.json 2> gpurun_out/r05_modes/run$i.err || exit 9
  python - gpurun_out/r05_modes/run$i.json <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print('value', d['value'], d['ms_per_step'])
PY
  python tools/r05/chain_summary.py gpurun_out/r05_modes/run$i.err
  rm gpurun_out/r05_modes/run$i.err
done
