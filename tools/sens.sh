#!/bin/bash
# resource sensitivity of the 16-wave kernel (diagnostic -DSSQ_SENS build): time versus extra VALU instructions
# (low 16 bits of SSQ_ABLATE) and extra LDS reads (high 16 bits) per frame.  One device, interleaved.
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
export SSQ_HIP_LIB=$ROOT/ssqueeze_rs_amd/libssq_hip_sens.so
for r in 1 2; do
  for A in 0 200 400 800 $((32<<16)) $((64<<16)) $((128<<16)); do
    SSQ_ABLATE=$A python bench.py --steps 10 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "import sys,json;d=json.loads(sys.stdin.read());print('round $r extra_valu=$((A & 65535)) extra_lds=$((A >> 16))', round(d['ms_per_step'],4), 'ms')"
  done
done
