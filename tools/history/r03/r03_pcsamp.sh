#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/r03pc
mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
M=${1:-stochastic}; U=${2:-cycles}; I=${3:-1048576}
timeout -k 10 400 rocprofv3 --kernel-trace --pc-sampling-beta-enabled --pc-sampling-method $M --pc-sampling-unit $U --pc-sampling-interval $I --output-format csv -d $OUT/$M -- python3 $R/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-host-leg --no-api-leg --no-configs4 --no-profile --contexts 1 > $OUT/$M.json 2> $OUT/$M.err
echo "rc=$?"; tail -5 $OUT/$M.err; ls -la $OUT/$M/* | head -20
