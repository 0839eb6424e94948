#!/bin/bash
# The long randomised parity sweep + determinism soak on the final code of round 5 (prints a progress line per seed).
OUT=gpurun_out/r05_robust; mkdir -p $OUT
for s in ${SEEDS:-21 22 23 24 25 26}; do
  timeout -k 10 400 python tools/fuzz_parity.py ${FUZZ:-500} $s > $OUT/fuzz_seed$s.log 2>&1; echo "fuzz seed $s: $(tail -1 $OUT/fuzz_seed$s.log)"
done
timeout -k 10 300 python tools/soak.py 3000 > $OUT/soak_long.log 2>&1; echo "soak: $(tail -1 $OUT/soak_long.log)"
timeout -k 10 300 python tools/soak.py 3000 host > $OUT/soak_long_host.log 2>&1; echo "soak host: $(tail -1 $OUT/soak_long_host.log)"
