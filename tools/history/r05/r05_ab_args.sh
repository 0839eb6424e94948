#!/bin/bash
# Same-call A/B of bench.py argument sets:  tools/r05/r05_ab_args.sh TAG "args ..." "args ..." ...   ("-" = none).
# Each setting: bench.py --steps 200 without the side legs, twice, alternating; prints value, ms/step and the per-kernel sums.
TAG=$1; shift
mkdir -p gpurun_out
for rnd in 1 2; do
for v in "$@"; do
  name=$(echo "$v" | tr ' =/-' '____' | tail -c 40)
  if [ "$v" = "-" ]; then extra=""; else extra="$v"; fi
  timeout -k 10 300 python bench.py --steps 200 --no-cpu-baseline --no-api-leg --no-configs4 --no-host-leg $extra > gpurun_out/${TAG}_$name.json 2> gpurun_out/${TAG}_$name.err || { echo "bench $v failed"; tail -3 gpurun_out/${TAG}_$name.err; exit 9; }
  python - "$v" gpurun_out/${TAG}_$name.json <<'PY'
import json,sys
d=json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
print(sys.argv[1], d["value"], "ms/step", d["ms_per_step"], {k:round(v,3) for k,v in d.get("kernel_ms_per_step",{}).items()})
PY
done
done
