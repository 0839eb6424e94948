#!/bin/bash
# Pipelined rate by (contexts, GPU_MAX_HW_QUEUES of the HIP runtime), interleaved repetitions, same call.
mkdir -p gpurun_out
for rnd in 1 2 3; do
CQ=${CQ:-6:4 8:5 8:4 10:6 7:5 9:5 12:5 8:6}
for cq in $CQ; do
  c=${cq%%:*}; q=${cq##*:}
  GPU_MAX_HW_QUEUES=$q timeout -k 10 200 python bench.py --steps 200 --contexts $c --no-cpu-baseline --no-api-leg --no-configs4 --no-host-leg --no-steady > gpurun_out/r05_cq.json 2> gpurun_out/r05_cq.err || { echo "bench $cq failed"; tail -3 gpurun_out/r05_cq.err; exit 9; }
  python - $c $q gpurun_out/r05_cq.json <<'PY'
import json,sys
d=json.loads(open(sys.argv[3]).read().strip().splitlines()[-1])
print(f"contexts {sys.argv[1]} queues {sys.argv[2]}: {d['value']:.0f} Mpix/s  {d['ms_per_step']} ms/step")
PY
done
done
