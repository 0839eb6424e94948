#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/r03bind
mkdir -p $OUT; cd $R
nproc; cat /sys/devices/system/node/online 2>/dev/null; python - <<'PY'
import sys
sys.path.insert(0,'.')
from hessgpu_amd import numa
print("pci", numa.gpu_pci_addresses()[:2], "local cpus of gpu 0:", len(numa.local_cpus(0) or []))
PY
for rep in 1 2; do
for b in 0 1; do
  HESS_BENCH_BIND=$b timeout -k 10 300 python bench.py --steps 100 --no-cpu-baseline --no-api-leg --no-configs4 --no-profile > $OUT/b$b_$rep.json 2>/dev/null
  python - <<PY
import json
d=json.loads(open("$OUT/b$b_$rep.json").read().strip().splitlines()[-1]); print("bind=$b", d["value"], d["value_host_to_host"], d["latency_ms_single_image"], d["config"].get("cpu_binding"))
PY
done
done
