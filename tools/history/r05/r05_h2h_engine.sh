#!/bin/bash
# host-to-host rate with the upload left to ROCr (HESS_UPLOAD_ENGINE=0) against an engine of the preferred host->device set
for rnd in 1 2 3 4; do
for v in "HESS_UPLOAD_ENGINE=0" "HESS_X=1"; do
  env $v python bench.py --steps 200 --no-cpu-baseline --no-api-leg --no-configs4 --no-profile 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v: device-resident', d['value'], ' host-to-host', d['value_host_to_host'])"
done
done
