#!/bin/bash
# the small octaves by one chain launch per octave from octave N on (HESS_CHAIN_FROM, developer build), pipelined line, twice
R=${GRAFT_REPO_ROOT:-$PWD}
export HESS_LIB=$R/hessgpu_amd/dev/libhessgpu.so
for rnd in 1 2; do for cf in 0 3 4 5; do
  HESS_CHAIN_FROM=$cf python bench.py --steps 200 --no-cpu-baseline --no-api-leg --no-configs4 --no-host-leg --no-real-images --no-matcher 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('HESS_CHAIN_FROM=$cf', d['value'], 'ms/step', d['ms_per_step'], {k:round(v,3) for k,v in d.get('kernel_ms_per_step',{}).items()})"
done; done
