#!/bin/bash
# NOTE (round 3): the SSQ_CWT_* tuning switches exist only in -DSSQ_TUNING builds: python -m ssqueeze_rs_amd.build --tune, then SSQ_HIP_LIB=$PWD/ssqueeze_rs_amd/libssq_hip_tune.so
# kernel-time breakdown of the CWT bench for each ablation library: tools/abl_cwt.sh <suffix>...
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
for S in "$@"; do
  if [ "$S" = base ]; then unset SSQ_HIP_LIB; else export SSQ_HIP_LIB=$ROOT/ssqueeze_rs_amd/libssq_hip_$S.so; fi
  rocprofv3 --kernel-trace --stats -d $ROOT/gpurun_out/abl_$S -o cwt -- python3 $ROOT/tools/bench_cwt.py --steps 2 > $ROOT/gpurun_out/abl_$S.log 2>&1
  echo "== $S"; python3 $ROOT/tools/rocpd_stats.py $ROOT/gpurun_out/abl_$S/cwt_results.db 3 | head -8
done
