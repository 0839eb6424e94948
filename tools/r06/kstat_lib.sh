#!/bin/bash
# per-kernel statistics (one stream) for library variants: tools/r06/kstat_lib.sh TAG kernel-substring lib lib ...   ("cur" = in-tree)
TAG=$1; PAT=$2; shift; shift
R=${GRAFT_REPO_ROOT:-$PWD}
cd /tmp && export TMPDIR=/tmp
for v in "$@"; do
  OUT=$R/gpurun_out/$TAG/$v; mkdir -p $OUT
  if [ "$v" = "cur" ]; then unset HESS_LIB; else export HESS_LIB=$R/tools/_variants/$v/libhessgpu.so; fi
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-steady --no-host-leg --no-api-leg --no-configs4 --contexts 1 > $OUT/bench.json 2> $OUT/bench.err
  find $OUT -name '*kernel_trace.csv' -delete; find $OUT -name '*agent_info.csv' -delete
  python3 - $OUT $v "$PAT" <<'PY'
import csv,glob,sys
for f in sorted(glob.glob(sys.argv[1]+"/**/*kernel_stats.csv", recursive=True)):
    for r in csv.DictReader(open(f)):
        n=r['Name'].replace('hess::(anonymous namespace)::','').replace('void ','')
        if any(p in n for p in sys.argv[3].split(',')):
            print(f"{sys.argv[2]:14s} {n[:40]:40s} calls {r['Calls']:>5s} avg_us {float(r['AverageNs'])/1e3:8.1f}")
PY
done
