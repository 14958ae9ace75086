#!/bin/bash
# NOTE (round 3): the SSQ_CWT_* tuning switches exist only in -DSSQ_TUNING builds: python -m ssqueeze_rs_amd.build --tune, then SSQ_HIP_LIB=$PWD/ssqueeze_rs_amd/libssq_hip_tune.so
# interleaved A/B of an environment switch on the ssq_cwt C4 bench, one device: tools/ab_cwt_env.sh VAR v1 v2 ...
VAR=$1; shift
for r in 1 2 3; do
  for V in "$@"; do
    env $VAR=$V python tools/bench_cwt.py --steps 5 ${CWT_ARGS:-} 2>/dev/null | python -c "import sys,json;d=json.loads(sys.stdin.read());print('round $r', '$VAR=$V', round(d['ms'],3), 'ms', round(d['frac_of_8TBps'],4))"
  done
done
