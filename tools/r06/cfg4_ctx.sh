#!/bin/bash
for rnd in 1 2; do for n in 2 3 4 5 6; do
  timeout -k 10 200 python tools/r06/cfg4_probe.py --contexts $n --runs 3 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('contexts', $n, d['ms_three_contexts'], 'ms/img', d['Mpix_per_s_three_contexts'], 'Mpix/s')"
done; done
