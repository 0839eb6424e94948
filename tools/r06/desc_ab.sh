#!/bin/bash
# descriptor kernel of library variants, same call: tools/r06/desc_ab.sh TAG lib lib ...   ("cur" = in-tree)
#   1080p bench workload on one stream (two launches of four images per step) and configs[4] (--delivery dma: four launches
#   over quarters of the list), average launch time by rocprofv3 --kernel-trace --stats
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-$PWD}
cd /tmp && export TMPDIR=/tmp
for v in "$@"; do
  OUT=$R/gpurun_out/$TAG/$v; mkdir -p $OUT
  if [ "$v" = "cur" ]; then unset HESS_LIB; else export HESS_LIB=$R/tools/_variants/$v/libhessgpu.so; fi
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-steady --no-host-leg --no-api-leg --no-configs4 --no-real-images --no-matcher --contexts 1 > $OUT/bench.json 2> $OUT/bench.err || { tail -3 $OUT/bench.err; exit 3; }
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats4 -- python3 $R/tools/r06/cfg4_probe.py --quick --runs 10 --delivery dma > $OUT/cfg4.json 2> $OUT/cfg4.err || { tail -3 $OUT/cfg4.err; exit 4; }
  find $OUT -name '*kernel_trace.csv' -delete; find $OUT -name '*agent_info.csv' -delete
  python3 - $OUT $v <<'PY'
import csv,glob,sys
for sub in ("stats","stats4"):
    for f in sorted(glob.glob(sys.argv[1]+"/"+sub+"/**/*kernel_stats.csv", recursive=True)):
        for r in csv.DictReader(open(f)):
            n=r['Name'].replace('hess::(anonymous namespace)::','').replace('void ','')
            if 'descriptor_pixel' in n:
                print(f"{sys.argv[2]:12s} {'1080p x4' if sub=='stats' else 'configs[4]/4':12s} {n[:30]:30s} calls {r['Calls']:>5s} avg_us {float(r['AverageNs'])/1e3:8.1f}")
PY
done
