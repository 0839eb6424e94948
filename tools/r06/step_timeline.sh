#!/bin/bash
# the launches of ONE step of eight 1080p images on one stream, in order, with their durations: tools/r06/step_timeline.sh [TAG] [lib]
R=${GRAFT_REPO_ROOT:-$PWD}; OUT=$R/gpurun_out/${1:-r06_step}; mkdir -p $OUT
if [ -n "$2" ] && [ "$2" != "cur" ]; then export HESS_LIB=$R/tools/_variants/$2/libhessgpu.so; fi
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $OUT/trace -- python3 $R/tools/single_image_trace.py 8 > /dev/null 2> $OUT/err.txt || { tail -3 $OUT/err.txt; exit 3; }
python3 - $OUT <<'PY'
import csv,glob,sys
t=glob.glob(sys.argv[1]+"/**/*kernel_trace.csv", recursive=True)[0]
rows=sorted(csv.DictReader(open(t)), key=lambda r:int(r['Start_Timestamp']))
rows=[r for r in rows if 'hess' in r['Kernel_Name']]
# steps end with the second descriptor launch; take the last complete step
ends=[i for i,r in enumerate(rows) if 'descriptor' in r['Kernel_Name']]
last=ends[-1]; first=ends[-3]+1
t0=int(rows[first]['Start_Timestamp']); tot=0
for r in rows[first:last+1]:
    n=r['Kernel_Name'].replace('hess::(anonymous namespace)::','').replace('void ','').split('(')[0]
    d=(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3; tot+=d
    print(f"  {(int(r['Start_Timestamp'])-t0)/1e3:8.1f} us  +{d:7.1f}  {n[:34]:34s} grid {int(r['Grid_Size_X'])//int(r['Workgroup_Size_X']):6d} x {r['Grid_Size_Y']}")
print("  sum of kernel time (us):", round(tot,1), " step (first start .. last end):", round((int(rows[last]['End_Timestamp'])-t0)/1e3,1))
PY
rm -rf $OUT/trace
