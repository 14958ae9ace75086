ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
for S in base cabl1 cabl2 cabl4 cabl7; do
  if [ "$S" = base ]; then unset SSQ_HIP_LIB; else export SSQ_HIP_LIB=$ROOT/ssqueeze_rs_amd/libssq_hip_$S.so; fi
  rocprofv3 --kernel-trace --stats -d $ROOT/gpurun_out/ablc5_$S -o cwt -- python3 $ROOT/tools/bench_cwt.py --dtype f64 --log2n 22 --steps 1 > $ROOT/gpurun_out/ablc5_$S.log 2>&1
  echo "== $S"; python3 $ROOT/tools/rocpd_stats.py $ROOT/gpurun_out/ablc5_$S/cwt_results.db | head -5
done
