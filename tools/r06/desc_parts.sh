#!/bin/bash
# descriptor launches / result transfers per batch of eight (HESS_DESC_PARTS, developer build), pipelined line, twice
R=${GRAFT_REPO_ROOT:-$PWD}
export HESS_LIB=$R/hessgpu_amd/dev/libhessgpu.so
for rnd in 1 2; do for n in 0 1 2 3 4; do
  HESS_DESC_PARTS=$n python bench.py --steps 200 --no-cpu-baseline --no-api-leg --no-configs4 --no-host-leg --no-real-images --no-matcher --no-steady 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('HESS_DESC_PARTS=$n', d['value'], 'ms/step', d['ms_per_step'], {k:round(v,3) for k,v in d.get('kernel_ms_per_step',{}).items()})"
done; done
