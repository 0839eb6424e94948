#!/bin/bash
# GPU-side picture of the driver's 20 timed steps: kernel trace of bench.py --steps 20 --warmup 5 (no legs after the region),
# the region = the launches after the last fence; busy time, kernels in flight and launches per hardware queue in 0.5 ms slices
R=${GRAFT_REPO_ROOT:-$PWD}; OUT=$R/gpurun_out/${1:-r06_region}; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $OUT/trace -- python3 $R/bench.py --steps ${2:-20} --warmup 5 --no-cpu-baseline --no-api-leg --no-host-leg --no-configs4 --no-real-images --no-matcher --no-steady --no-profile > $OUT/bench.json 2> $OUT/err.txt || { tail -3 $OUT/err.txt; exit 3; }
python3 - $OUT <<'PY'
import csv,glob,sys,json,collections
out=sys.argv[1]
d=json.loads(open(out+"/bench.json").read().strip().splitlines()[-1]); print("bench line: value", d["value"], "ms/step", d["ms_per_step"], "-> region", round(d["ms_per_step"]*d["steps"],2), "ms")
t=glob.glob(out+"/trace/**/*kernel_trace.csv", recursive=True)[0]
rows=[r for r in csv.DictReader(open(t)) if 'hess' in r['Kernel_Name']]
rows.sort(key=lambda r:int(r['Start_Timestamp']))
S=[int(r['Start_Timestamp']) for r in rows]; E=[int(r['End_Timestamp']) for r in rows]
# last idle gap > 60 us (the fence before the timed region)
cut=0; maxend=E[0]
for i in range(1,len(rows)):
    if S[i]-maxend>60000: cut=i
    maxend=max(maxend,E[i])
reg=rows[cut:]; t0=S[cut]; t1=max(E[cut:])
print("region by the trace: launches", len(reg), "from first start to last end", round((t1-t0)/1e6,3), "ms")
q=collections.Counter(r['Queue_Id'] for r in reg); print("launches per hardware queue:", dict(q))
sl=500000
n=int((t1-t0)/sl)+1
busy=[0]*n; infl=[0.0]*n; perq=[collections.Counter() for _ in range(n)]
ev=[]
for r in reg:
    s,e=int(r['Start_Timestamp'])-t0,int(r['End_Timestamp'])-t0
    ev.append((s,1)); ev.append((e,-1))
    k=s//sl
    while k*sl<e:
        a=max(s,k*sl); b=min(e,(k+1)*sl); infl[k]+=(b-a); perq[k][r['Queue_Id']]+=(b-a); k+=1
ev.sort(); cur=0; last=0
for tt,dv in ev:
    if cur>0:
        k=last//sl
        while k*sl<tt:
            a=max(last,k*sl); b=min(tt,(k+1)*sl); busy[k]+=(b-a); k+=1
    cur+=dv; last=tt
for k in range(n):
    w=min(sl,(t1-t0)-k*sl)
    print(f"  {k*0.5:5.1f} ms  busy {busy[k]/w:4.2f}  in flight {infl[k]/w:4.2f}  per queue " + " ".join(f"{qq}:{perq[k][qq]/w:4.2f}" for qq in sorted(q)))
PY
rm -rf $OUT/trace
