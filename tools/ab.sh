#!/bin/bash
# interleaved A/B of library builds on ONE device: tools/ab.sh libA.so libB.so [...]  (3 rounds each)
for r in 1 2 3; do
  for L in "$@"; do
    SSQ_HIP_LIB=$PWD/ssqueeze_rs_amd/$L python bench.py --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "import sys,json;d=json.loads(sys.stdin.read());print('round $r', '$L', round(d['ms_per_step'],4), 'ms', round(d['roofline']['frac'],4))"
  done
done
