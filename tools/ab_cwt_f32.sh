#!/bin/bash
# interleaved A/B of library builds on BASELINE config 4 (ssq_cwt fp32, 1 x 2^20, 256 scales): tools/ab_cwt_f32.sh ROUNDS libA.so ...
R=$1; shift
for r in $(seq 1 $R); do
  for L in "$@"; do
    SSQ_HIP_LIB=$PWD/ssqueeze_rs_amd/$L python tools/bench_cwt.py --steps 10 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('round $r $L', round(d['ms'],4))"
  done
done
