#!/bin/bash
# Which engine moves a device->host copy (tools/micro/d2h_engine.hip), and the corrected VALU peak table.
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/r03
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for m in hip hipthr hsa hsaeng; do
  timeout -k 10 120 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/eng_$m -- $R/tools/micro/d2h_engine $m > $OUT/eng_$m.txt 2>&1
  echo "== $m rc=$?"; grep -v "^W2\|^E2\|rocprof" $OUT/eng_$m.txt | tail -8
  cat $OUT/eng_$m/*/*kernel_stats.csv | cut -c1-120
  find $OUT/eng_$m -name '*kernel_trace.csv' -delete
done
GPU_FORCE_BLIT_COPY_SIZE=0 timeout -k 10 60 $R/tools/micro/d2h_engine hip > $OUT/eng_hip_noblit.txt 2>&1; tail -6 $OUT/eng_hip_noblit.txt
AMD_LOG_LEVEL=4 timeout -k 10 60 $R/tools/micro/d2h_engine hip 2>&1 | grep -i "HSA Copy\|copy_engine\|Blit" | sed 's/0x[0-9a-f]*/X/g; s/[0-9]\{5,\}/N/g' | sort | uniq -c | sort -rn | head -10 > $OUT/eng_hip_log.txt; cat $OUT/eng_hip_log.txt
timeout -k 10 200 $R/tools/micro/valu_peak > $OUT/valu_peak.txt 2>&1; echo "valu_peak rc=$?"
