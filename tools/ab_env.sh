#!/bin/bash
# interleaved A/B of an environment switch on ONE device: tools/ab_env.sh VAR val1 val2 ...
VAR=$1; shift
for r in 1 2 3; do
  for V in "$@"; do
    env $VAR=$V python bench.py --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "import sys,json;d=json.loads(sys.stdin.read());print('round $r', '$VAR=$V', round(d['ms_per_step'],4), 'ms', round(d['roofline']['frac'],4))"
  done
done
