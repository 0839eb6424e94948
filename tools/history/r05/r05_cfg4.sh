#!/bin/bash
# configs[4] (one 4096^2 image, -topk 65536 -half) by delivery form: tools/r05/r05_cfg4.sh "-" "HESS_DELIVERY=dma" ...
for rnd in 1 2; do
for v in "$@"; do
  if [ "$v" = "-" ]; then envs=""; else envs="$v"; fi
  env $envs timeout -k 10 300 python bench.py --steps 20 --no-cpu-baseline --no-api-leg --no-host-leg --no-steady > gpurun_out/r05_cfg4.json 2> gpurun_out/r05_cfg4.err || { echo "failed $v"; tail -3 gpurun_out/r05_cfg4.err; exit 9; }
  python - "$v" <<'PY'
import json,sys
d=json.loads(open('gpurun_out/r05_cfg4.json').read().strip().splitlines()[-1])['configs4']
print(sys.argv[1], 'one ctx', d['ms_per_image_one_context'], 'ms  three ctx', d['Mpix_per_s_three_contexts'], 'Mpix/s', d['kernel_ms_per_image'], d['roofline_descriptor']['kernel'][:34])
PY
done
done
