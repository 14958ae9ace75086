set -e
cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests -x -q -m gpu 2>&1 | tail -3
timeout -k 10 280 python bench.py > gpurun_out/bench_final.json 2> gpurun_out/bench_final.err
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/trace_final -- python3 $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-secondary > $GRAFT_REPO_ROOT/gpurun_out/bench_profiled.json 2>/dev/null
cd $GRAFT_REPO_ROOT
python tools/trace_timed.py gpurun_out/trace_final gpurun_out/bench_profiled.json > gpurun_out/trace_timed.json
rm -rf gpurun_out/trace_final
