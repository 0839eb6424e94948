# EXPERIMENT: what would a pipeline without materialised gradient planes cost / gain?
#   A  HESS_X_NOGOT_AFTER=3: the Gaussian launches stop computing + writing (gradient, theta) after three batches per context
#   B  the same + a descriptor kernel that does two more gathers and a sqrt + atan2 per sample (xdesc variant)
R=$PWD
for rnd in 1 2; do
  for v in "base" "A" "B" "Bonly"; do
    unset HESS_LIB HESS_X_NOGOT_AFTER
    case $v in
      A) export HESS_X_NOGOT_AFTER=3;;
      B) export HESS_X_NOGOT_AFTER=3; export HESS_LIB=$R/tools/_variants/xdesc/libhessgpu.so;;
      Bonly) export HESS_LIB=$R/tools/_variants/xdesc/libhessgpu.so;;
    esac
    python bench.py --steps 200 --no-cpu-baseline --no-api-leg --no-configs4 --no-host-leg 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v', d['value'], 'ms/step', d['ms_per_step'], {k:round(x,3) for k,x in d['kernel_ms_per_step'].items()})"
  done
done
