#!/bin/bash
# bench.py's Python step loop against the same loop in C++ (apps/pipeline.cpp), same box, same call, 200 steps each, twice.
R=${GRAFT_REPO_ROOT:-$PWD}
mkdir -p /tmp/px && python - <<'PY'
import sys
sys.path[:0] = ["tests"]
import fixtures
for i in range(8):
    im = fixtures.synthetic_blobs(1920, 1080, i)
    open(f"/tmp/px/{i}.pgm", "wb").write(b"P5\n1920 1080\n255\n" + im.tobytes())
PY
ARGS=""; for i in 0 1 2 3 4 5 6 7; do ARGS="$ARGS -i /tmp/px/$i.pgm"; done
for rnd in 1 2; do
  for c in ${CONTEXTS:-7}; do
    python bench.py --steps 200 --contexts $c --no-cpu-baseline --no-api-leg --no-configs4 --no-host-leg --no-profile | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('python loop, contexts', d['config']['pipelined_contexts_per_gpu'], d['value'], 'Mpix/s', d['ms_per_step'], 'ms/step')"
    $R/hessgpu_amd/bin/pipeline $ARGS -n 200 -contexts $c
  done
done
