#!/bin/bash
# parity + a short bench line:  tools/r06/check.sh TAG [pytest args]
TAG=${1:-chk}; shift
mkdir -p gpurun_out/$TAG
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_keypoint_list_gpu.py tests/test_reference_inputs_gpu.py -q -m gpu -x "$@" > gpurun_out/$TAG/parity.txt 2>&1
echo "parity rc=$?" >> gpurun_out/$TAG/parity.txt
tail -5 gpurun_out/$TAG/parity.txt
grep -q "parity rc=0" gpurun_out/$TAG/parity.txt || exit 1
for k in 1 2; do
timeout -k 10 300 python bench.py --steps 200 --no-cpu-baseline --no-api-leg --no-host-leg > gpurun_out/$TAG/bench_$k.json 2> gpurun_out/$TAG/bench_$k.err || { tail -5 gpurun_out/$TAG/bench_$k.err; exit 2; }
python - gpurun_out/$TAG/bench_$k.json <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
c4=d.get("configs4",{})
print("value", d["value"], "ms/step", d["ms_per_step"], d.get("kernel_ms_per_step"), "| cfg4", c4.get("Mpix_per_s_one_context"), c4.get("Mpix_per_s_three_contexts"), c4.get("kernel_ms_per_image"))
PY
done
