#!/bin/bash
# as region_timeline.sh, but step by step: for every step of the timed region (a stream's run of launches from a gauss_first
# to its second descriptor launch) the hardware queue, first start, last end, and kernel time
R=${GRAFT_REPO_ROOT:-$PWD}; OUT=$R/gpurun_out/${1:-r06_region}; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $OUT/trace -- python3 $R/bench.py --steps ${2:-20} --warmup 5 --no-cpu-baseline --no-api-leg --no-host-leg --no-configs4 --no-real-images --no-matcher --no-steady --no-profile > $OUT/bench.json 2> $OUT/err.txt || { tail -3 $OUT/err.txt; exit 3; }
python3 - $OUT <<'PY'
import csv,glob,sys,json,collections
out=sys.argv[1]
d=json.loads(open(out+"/bench.json").read().strip().splitlines()[-1]); print("bench line: value", d["value"], "ms/step", d["ms_per_step"], "-> region", round(d["ms_per_step"]*d["steps"],2), "ms")
t=glob.glob(out+"/trace/**/*kernel_trace.csv", recursive=True)[0]
allrows=list(csv.DictReader(open(t)))
print("columns:", list(allrows[0].keys()))
rows=[r for r in allrows if 'hess' in r['Kernel_Name']]
rows.sort(key=lambda r:int(r['Start_Timestamp']))
S=[int(r['Start_Timestamp']) for r in rows]; E=[int(r['End_Timestamp']) for r in rows]
cut=0; maxend=E[0]
for i in range(1,len(rows)):
    if S[i]-maxend>60000: cut=i
    maxend=max(maxend,E[i])
reg=rows[cut:]; t0=S[cut]
key='Stream_Id' if 'Stream_Id' in reg[0] else 'Thread_Id'
by=collections.defaultdict(list)
for r in reg: by[r[key]].append(r)
steps=[]
for sid,rs in by.items():
    cur=[]
    for r in rs:
        cur.append(r)
        if 'descriptor' in r['Kernel_Name'] and sum('descriptor' in x['Kernel_Name'] for x in cur)==2:
            steps.append((sid,cur)); cur=[]
    if cur: steps.append((sid,cur))
steps.sort(key=lambda x:int(x[1][0]['Start_Timestamp']))
for sid,rs in steps:
    s=int(rs[0]['Start_Timestamp'])-t0; e=max(int(x['End_Timestamp']) for x in rs)-t0
    kt=sum(int(x['End_Timestamp'])-int(x['Start_Timestamp']) for x in rs)
    print(f"  {key} {sid:>6s} queue {rs[0]['Queue_Id']}  launches {len(rs):3d}  start {s/1e6:7.3f}  end {e/1e6:7.3f}  span {(e-s)/1e6:6.3f}  kernel time {kt/1e6:6.3f} ms")
PY
rm -rf $OUT/trace
