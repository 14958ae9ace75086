#!/bin/bash
# timing-only ablations of the fused kernel (results are wrong by design); prints ms/step per mask
for m in 0 1 2 4 8 16 32 64 96 3 127; do
  SSQ_ABLATE=$m python bench.py --batch 64 --steps 10 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('ablate=$m', round(d['ms_per_step'],4), 'ms')"
done
