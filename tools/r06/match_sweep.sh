#!/bin/bash
# matcher: workgroup targets (HESS_MATCH_WGS), twice
for rnd in 1 2; do
for w in 256 384 512 768; do
  echo "== wgs $w"; HESS_MATCH_WGS=$w timeout -k 10 120 python tools/bench_match.py | grep -o '"n1": [0-9]*, "n2": [0-9]*, "device_ms": [0-9.]*, "GMAC_per_s": [0-9.]*' | tail -2
done
done
