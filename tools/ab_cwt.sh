#!/bin/bash
# NOTE (round 3): the SSQ_CWT_* tuning switches exist only in -DSSQ_TUNING builds: python -m ssqueeze_rs_amd.build --tune, then SSQ_HIP_LIB=$PWD/ssqueeze_rs_amd/libssq_hip_tune.so
# interleaved A/B of library variants / env switches for the CWT bench on ONE device
# usage: tools/ab_cwt.sh "ENV1=.. ENV2=.." "ENV=.." ...   (each argument is one configuration's environment)
for r in 1 2; do
  for CFG in "$@"; do
    env $CFG python tools/bench_cwt.py --steps 5 2>/dev/null | python -c "import sys,json;d=json.loads(sys.stdin.read());print('round $r', '$CFG', round(d['ms'],3), 'ms')"
  done
done
