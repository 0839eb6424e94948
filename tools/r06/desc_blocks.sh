#!/bin/bash
# descriptor launch grid (HESS_DESC_MAX_BLOCKS) on configs[4], synchronous run (one launch) and copier delivery (four launches)
mkdir -p gpurun_out/r06_desc
for rnd in 1 2; do
for mb in 2048 4096 8192 16384; do
  for dl in mirror dma; do
    HESS_DESC_MAX_BLOCKS=$mb timeout -k 10 200 python tools/r06/cfg4_probe.py --delivery $dl > gpurun_out/r06_desc/blocks_${mb}_${dl}_$rnd.json 2> gpurun_out/r06_desc/blocks_${mb}_${dl}_$rnd.err || { echo failed $mb $dl; tail -3 gpurun_out/r06_desc/blocks_${mb}_${dl}_$rnd.err; exit 9; }
    python - $mb $dl gpurun_out/r06_desc/blocks_${mb}_${dl}_$rnd.json <<'PY'
import json,sys
d=json.loads(open(sys.argv[3]).read().strip().splitlines()[-1])
print("max_blocks", sys.argv[1], sys.argv[2], "desc ms", d["kernel_ms_per_image"]["descriptor"], "launches", d["descriptor_launches_per_image"], "one ctx", d["ms_one_context"], "three ctx", d["ms_three_contexts"], d["Mpix_per_s_three_contexts"])
PY
  done
done
done
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r06_desc/blocks_2048_dma_1.json").read().strip().splitlines()[-1])
print(d["footprint"], d["kernel_ms_per_image"])
PY
