#!/bin/bash
# more randomised parity cases on the final code (two seeds), each bounded well below the 1200 s call limit
OUT=gpurun_out/r05_robust; mkdir -p $OUT
timeout -k 10 500 python tools/fuzz_parity.py ${FUZZ:-250} 11 > $OUT/fuzz_seed11.log 2>&1; echo "fuzz seed 11: $(tail -1 $OUT/fuzz_seed11.log)"
timeout -k 10 500 python tools/fuzz_parity.py ${FUZZ:-250} 12 > $OUT/fuzz_seed12.log 2>&1; echo "fuzz seed 12: $(tail -1 $OUT/fuzz_seed12.log)"
