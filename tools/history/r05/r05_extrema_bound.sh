#!/bin/bash
# What is the extrema scan's 3.5 TB/s made of?  The access-pattern microbenchmark with the caches
# evicted by a FILL (dirty lines, round 4's form) and by a READ pass (clean lines), each also under rocprofv3 with
# WRITE_SIZE / FETCH_SIZE per kernel (separate passes).
R=${GRAFT_REPO_ROOT:-$PWD}; OUT=$R/gpurun_out/${1:-r05_extrema}; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for mode in fill read; do
  $R/tools/micro/strided_streams $mode > $OUT/strided_$mode.txt 2>&1
  cat $OUT/strided_$mode.txt
  for ctr in WRITE_SIZE FETCH_SIZE; do
    rm -rf /tmp/ss_${mode}_$ctr
    rocprofv3 --kernel-trace --pmc $ctr --output-format csv -d /tmp/ss_${mode}_$ctr -- $R/tools/micro/strided_streams $mode > /dev/null 2>&1
    python3 - $(find /tmp/ss_${mode}_$ctr -name '*counter_collection.csv' | head -1) $ctr $mode <<'PY'
import csv, sys, collections, re
tot, n = collections.Counter(), collections.Counter()
for r in csv.DictReader(open(sys.argv[1])):
    if r["Counter_Name"] != sys.argv[2]: continue
    k = re.sub(r"\(.*", "", r["Kernel_Name"].replace("void ", ""))
    tot[k] += float(r["Counter_Value"]); n[k] += 1
for k in tot: print(f"  [{sys.argv[3]}] {sys.argv[2]} per launch of {k[:40]:40s}: {tot[k] / n[k] / 1024:9.1f} MiB ({n[k]} launches)")
PY
  done
done
