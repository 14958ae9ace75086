#!/bin/bash
# interleaved A/B of library builds on the fp64 ssq_stft leg (batch 64): tools/ab_f64.sh ROUNDS libA.so libB.so ...
R=$1; shift
for r in $(seq 1 $R); do
  for L in "$@"; do
    SSQ_HIP_LIB=$PWD/ssqueeze_rs_amd/$L python tools/bench_stft.py --dtype f64 --batch 64 --steps 10 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('round $r $L', round(d['ms'],4))"
  done
done
