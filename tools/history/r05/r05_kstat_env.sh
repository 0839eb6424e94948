#!/bin/bash
# rocprofv3 kernel statistics (single stream, batch of 8) under environment settings, same call: tools/r05/r05_kstat_env.sh "VAR=v" ... ("-" = none)
R=${GRAFT_REPO_ROOT:-$PWD}; cd /tmp && export TMPDIR=/tmp
for v in "$@"; do
  n=$(echo "$v" | tr ' =/' '___')
  rm -rf /tmp/kstat_$n
  if [ "$v" = "-" ]; then unset HESS_NO_FIRST_FUSION HESS_NO_TOP_FUSION; else export $v; fi
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kstat_$n -- python3 $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-steady --no-host-leg --no-api-leg --no-configs4 --no-profile --contexts 1 > /dev/null 2>&1
  if [ "$v" != "-" ]; then unset ${v%%=*}; fi
  echo "== $v"; python3 - $(find /tmp/kstat_$n -name '*kernel_stats.csv') <<'PY'
import csv,sys,re
for r in list(csv.DictReader(open(sys.argv[1])))[:16]:
    n=re.sub(r"\(.*","",r["Name"].replace("(anonymous namespace)::","").replace("void ","").replace("hess::",""))
    print(f'  {n[:40]:40s} calls {int(r["Calls"]):5d} avg {float(r["AverageNs"])/1e3:8.1f} us  max {float(r["MaxNs"])/1e3:8.1f} {float(r["Percentage"]):5.1f}%')
PY
done
