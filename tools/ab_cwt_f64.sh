#!/bin/bash
# interleaved A/B of library builds on the fp64 ssq_cwt legs (C4 fp64 and one C5 signal): tools/ab_cwt_f64.sh ROUNDS libA.so ...
R=$1; shift
for r in $(seq 1 $R); do
  for L in "$@"; do
    for n in 20 22; do
      SSQ_HIP_LIB=$PWD/ssqueeze_rs_amd/$L python tools/bench_cwt.py --dtype f64 --log2n $n --steps 3 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('round $r $L 2^$n', round(d['ms'],3))"
    done
  done
done
