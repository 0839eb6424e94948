#!/bin/bash
# extrema_place_kernel of library variants: single-stream launch time (kstat) and the pipelined line: tools/r06/place_ab.sh TAG lib ...
TAG=$1; shift
bash tools/r06/kstat_lib.sh $TAG extrema_place "$@" && bash tools/r06/ab_lib.sh ${TAG}_ab "$@"
