#!/bin/bash
# bench.py over (batch, contexts) pairs on the GPU box:  tools/sweep_bench.sh TAG "B:C B:C ..."
TAG=$1; shift
for bc in $1; do
  b=${bc%%:*}; c=${bc##*:}
  timeout -k 10 300 python bench.py --steps 30 --batch $b --contexts $c --no-cpu-baseline --no-profile > gpurun_out/${TAG}_b${b}_c${c}.json 2> gpurun_out/${TAG}_b${b}_c${c}.err || { echo "bench $bc failed"; tail -3 gpurun_out/${TAG}_b${b}_c${c}.err; exit 9; }
  python - $b $c gpurun_out/${TAG}_b${b}_c${c}.json <<'PY'
import json,sys
d=json.loads(open(sys.argv[3]).read().strip().splitlines()[-1])
print("batch",sys.argv[1],"contexts",sys.argv[2],"value",d["value"],"h2h",d.get("value_host_to_host"),"lat",d.get("latency_ms_single_image"))
PY
done
