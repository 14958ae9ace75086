#!/bin/bash
# per-kernel times of the register-core CWT kernels under ablation builds (python -m ssqueeze_rs_amd.build --variant
# regablN -DSSQ_REG_ABL=N beforehand): tools/abl_cwt_reg.sh 0 1 2 4 ...
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
for A in "$@"; do
  if [ "$A" = 0 ]; then export SSQ_HIP_LIB=$ROOT/ssqueeze_rs_amd/libssq_hip.so; else export SSQ_HIP_LIB=$ROOT/ssqueeze_rs_amd/libssq_hip_regabl$A.so; fi
  rm -rf /tmp/prof_abl
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_abl -- python3 $ROOT/tools/bench_cwt.py --steps 3 > /tmp/prof_abl.log 2>&1
  f=$(find /tmp/prof_abl -name "*kernel_stats.csv" | head -1)
  echo "abl=$A $(grep -E 'cwt_reg_r[12]' $f | awk -F, '{printf "%s avg %.1f us  ", substr($1,7,14), $4/1000}')"
done
