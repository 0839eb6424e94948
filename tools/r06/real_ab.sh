#!/bin/bash
# real-image leg of bench.py for library variants: tools/r06/real_ab.sh TAG lib lib ...
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-$PWD}
mkdir -p gpurun_out/$TAG
for rnd in 1 2; do
for v in "$@"; do
  if [ "$v" = "cur" ]; then unset HESS_LIB; else export HESS_LIB=$R/tools/_variants/$v/libhessgpu.so; fi
  timeout -k 10 300 python bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-api-leg --no-host-leg --no-configs4 --no-matcher --no-profile --no-steady > gpurun_out/$TAG/${v}_$rnd.json 2> gpurun_out/$TAG/${v}_$rnd.err || { echo "bench $v failed"; tail -5 gpurun_out/$TAG/${v}_$rnd.err; exit 9; }
  python - "$v" gpurun_out/$TAG/${v}_$rnd.json <<'PY'
import json,sys
d=json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])["real_images"]
for k,v in d.items():
    if isinstance(v,dict) and "Mpix_per_s_three_contexts" in v:
        print(f"{sys.argv[1]:8s} {k[:28]:28s} {v['Mpix_per_s_three_contexts']:9.1f} Mpix/s  {v['ms_per_step']:7.3f} ms/step  feat {v['features_per_image_mean']:7.1f}  {v['kernel_ms_per_step']}")
PY
done
done
