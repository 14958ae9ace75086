#!/bin/bash
# NOTE (round 3): the SSQ_CWT_* tuning switches exist only in -DSSQ_TUNING builds: python -m ssqueeze_rs_amd.build --tune, then SSQ_HIP_LIB=$PWD/ssqueeze_rs_amd/libssq_hip_tune.so
# interleaved A/B of two environment switches on the ssq_cwt C4 bench: tools/ab_cwt_env2.sh "A=1 B=2" "A=3 B=4" ...
for r in 1 2; do
  for V in "$@"; do
    env $V python tools/bench_cwt.py --steps 5 ${CWT_ARGS:-} 2>/dev/null | python -c "import sys,json;d=json.loads(sys.stdin.read());print('round $r', '$V', round(d['ms'],3), 'ms', round(d['frac_of_8TBps'],4))"
  done
done
