#!/bin/bash
# one 1080p image through one context (the latency settings): kernel statistics + the timeline of one image
R=${GRAFT_REPO_ROOT:-$PWD}; OUT=$R/gpurun_out/${1:-r06_single}; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $R/tools/single_image_trace.py 1 > /dev/null 2> $OUT/err.txt
python3 - $OUT <<'PY'
import csv,glob,sys
f=glob.glob(sys.argv[1]+"/**/*kernel_stats.csv", recursive=True)[0]
tot=0
for r in csv.DictReader(open(f)):
    n=r['Name'].replace('hess::(anonymous namespace)::','').replace('void ','')
    per=float(r['TotalDurationNs'])/30/1e3
    tot+=per
    print(f"  {n[:44]:44s} calls/img {int(r['Calls'])/30:5.1f} avg_us {float(r['AverageNs'])/1e3:7.1f} us/img {per:7.1f}")
print("  sum of kernel time per image (us):", round(tot,1))
t=glob.glob(sys.argv[1]+"/**/*kernel_trace.csv", recursive=True)[0]
rows=sorted(csv.DictReader(open(t)), key=lambda r:int(r['Start_Timestamp']))
# the last image: the final 15 launches
last=rows[-15:]
t0=int(last[0]['Start_Timestamp'])
for r in last:
    n=r['Kernel_Name'].replace('hess::(anonymous namespace)::','').replace('void ','').split('(')[0]
    print(f"  {(int(r['Start_Timestamp'])-t0)/1e3:8.1f} .. {(int(r['End_Timestamp'])-t0)/1e3:8.1f} us  {n[:40]}")
PY
find $OUT -name '*kernel_trace.csv' -delete; find $OUT -name '*agent_info.csv' -delete
