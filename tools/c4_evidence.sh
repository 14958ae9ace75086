set -e
cd $GRAFT_REPO_ROOT
bash tools/pmc_cwt.sh r03c4b > /dev/null 2>&1
python tools/traffic_cwt.py gpurun_out/pmc_r03c4b/pass4 gpurun_out/pmc_r03c4b/pass5 > gpurun_out/cwt_traffic_b.json
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_c4b -o c4 -- python3 $GRAFT_REPO_ROOT/tools/bench_cwt.py --steps 4 > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
python tools/rocpd_stats.py gpurun_out/prof_c4b/c4_results.db > gpurun_out/cwt_c4b_kernel_stats.txt 2>&1 || true
