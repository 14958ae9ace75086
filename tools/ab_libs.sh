#!/bin/bash
# Interleaved A/B of library builds on ONE device (one gpurun call): tools/ab_libs.sh ROUNDS libA.so libB.so ...
# Each round runs tools/ab_inproc.py once per library (fresh process, 3 x 10 timed steps inside); prints every line.
R=$1; shift
for r in $(seq 1 $R); do
  for L in "$@"; do
    SSQ_HIP_LIB=$PWD/ssqueeze_rs_amd/$L python tools/ab_inproc.py --rounds 3 --steps 10 ${AB_ARGS:-} - 2>/dev/null | grep variant | sed "s/^/round $r $L /"
  done
done
