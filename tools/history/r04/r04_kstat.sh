#!/bin/bash
# rocprofv3 kernel statistics (single stream, batch of 8) of one or more library builds, same call: tools/r04/r04_kstat.sh lib...
R=${GRAFT_REPO_ROOT:-$PWD}; cd /tmp && export TMPDIR=/tmp
for v in "$@"; do
  if [ "$v" = "cur" ]; then unset HESS_LIB; else export HESS_LIB=$R/tools/_variants/$v/libhessgpu.so; fi
  rm -rf /tmp/kstat_$v
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kstat_$v -- python3 $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-steady --no-host-leg --no-api-leg --no-configs4 --no-profile --contexts 1 > /dev/null 2>&1
  echo "== $v"; python3 - $(find /tmp/kstat_$v -name '*kernel_stats.csv') <<'PY'
import csv,sys,re
for r in list(csv.DictReader(open(sys.argv[1])))[:12]:
    n=re.sub(r"\(.*","",r["Name"].replace("(anonymous namespace)::","").replace("void ","").replace("hess::",""))
    print(f'  {n[:40]:40s} calls {int(r["Calls"]):5d} avg {float(r["AverageNs"])/1e3:8.1f} us {float(r["Percentage"]):5.1f}%')
PY
done
