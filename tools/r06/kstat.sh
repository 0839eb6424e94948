#!/bin/bash
# per-kernel statistics of the bench workload on one stream (and optionally configs[4]):  tools/r06/kstat.sh TAG [cfg4]
TAG=${1:-kstat}
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-steady --no-host-leg --no-api-leg --no-configs4 --contexts 1 > $OUT/bench.json 2> $OUT/bench.err
if [ "$2" = "cfg4" ]; then
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats4 -- python3 $R/tools/r06/cfg4_probe.py --quick --runs 10 > $OUT/cfg4.json 2> $OUT/cfg4.err
fi
find $OUT -name '*kernel_trace.csv' -delete; find $OUT -name '*agent_info.csv' -delete
python3 - $OUT <<'PY'
import csv,glob,sys
for f in sorted(glob.glob(sys.argv[1]+"/**/*kernel_stats.csv", recursive=True)):
    print(f)
    for r in csv.DictReader(open(f)):
        n=r['Name'].replace('hess::(anonymous namespace)::','').replace('void ','')
        print(f"  {n[:44]:44s} calls {r['Calls']:>5s} avg_us {float(r['AverageNs'])/1e3:8.1f} total_ms {float(r['TotalDurationNs'])/1e6:7.2f}")
PY
