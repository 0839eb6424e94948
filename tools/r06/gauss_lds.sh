#!/bin/bash
# Gaussian launches of library variants: average launch time (rocprofv3 --kernel-trace --stats, one stream) and the LDS counters
# of one counter pass:  tools/r06/gauss_lds.sh TAG kernel-substring lib lib ...   ("cur" = in-tree)
TAG=$1; PAT=$2; shift; shift
R=${GRAFT_REPO_ROOT:-$PWD}
cd /tmp && export TMPDIR=/tmp
A="--no-cpu-baseline --no-steady --no-host-leg --no-api-leg --no-configs4 --no-real-images --no-matcher --contexts 1"
for v in "$@"; do
  OUT=$R/gpurun_out/$TAG/$v; mkdir -p $OUT
  if [ "$v" = "cur" ]; then unset HESS_LIB; else export HESS_LIB=$R/tools/_variants/$v/libhessgpu.so; fi
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $R/bench.py --steps 10 --warmup 2 $A > $OUT/bench.json 2> $OUT/bench.err || { tail -3 $OUT/bench.err; exit 3; }
  rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD --output-format csv -d $OUT/pmc -- python3 $R/bench.py --steps 2 --warmup 1 --no-profile $A > /dev/null 2> $OUT/pmc.err || { tail -3 $OUT/pmc.err; exit 4; }
  python3 - $OUT $v "$PAT" <<'PY'
import csv,glob,sys,collections
out,v,pat=sys.argv[1:4]
avg={}
for f in glob.glob(out+"/stats/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        n=r['Name'].replace('hess::(anonymous namespace)::','').replace('void ','').split('(')[0]
        avg[n]=(float(r['AverageNs'])/1e3, int(r['Calls']))
acc=collections.defaultdict(lambda: collections.defaultdict(float)); cnt=collections.Counter()
for f in glob.glob(out+"/pmc/**/*counter_collection.csv", recursive=True):
    seen=set()
    for r in csv.DictReader(open(f)):
        n=r['Kernel_Name'].replace('hess::(anonymous namespace)::','').replace('void ','').split('(')[0]
        acc[n][r['Counter_Name']]+=float(r['Counter_Value'])
        key=(n,r['Dispatch_Id'])
        if key not in seen: seen.add(key); cnt[n]+=1
for n in sorted(avg):
    if any(p in n for p in pat.split(',')):
        a=acc.get(n,{}); k=max(cnt.get(n,1),1)
        print(f"{v:10s} {n[:34]:34s} avg_us {avg[n][0]:7.1f} LDS active/launch {a.get('SQ_LDS_IDX_ACTIVE',0)/k:10.3g} conflicts {a.get('SQ_LDS_BANK_CONFLICT',0)/k:10.3g} share {a.get('SQ_LDS_BANK_CONFLICT',0)/max(a.get('SQ_LDS_IDX_ACTIVE',1),1):4.2f}")
PY
  find $OUT -name '*kernel_trace.csv' -delete; find $OUT -name '*agent_info.csv' -delete; find $OUT -name '*counter_collection.csv' -delete
done
