#!/bin/bash
# HBM traffic of the fp64 ssq_stft pass (batch 256 x 2^20, n_fft 1024): FETCH_SIZE / WRITE_SIZE, one rocprofv3 run each.
# usage (GPU box): bash tools/pmc_f64.sh  -> gpurun_out/pmc_f64/summary.txt
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/pmc_f64
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for P in FETCH_SIZE WRITE_SIZE; do
  i=$((i+1))
  rocprofv3 --pmc $P --output-format csv -d $OUT/pass$i -- python3 $ROOT/tools/bench_stft.py --dtype f64 --batch 256 --steps 2 > $OUT/pass$i.log 2>&1 || echo "pass $i failed"
done
python3 - <<PY
import csv, glob, collections
agg = collections.defaultdict(list)
for f in glob.glob("$OUT/pass*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "stft_fused" in r["Kernel_Name"]:
            agg[(r["Kernel_Name"][:70], r["Counter_Name"])].append(float(r["Counter_Value"]))
with open("$OUT/summary.txt", "w") as o:
    for (k, c), v in sorted(agg.items()):
        line = f"{k:70s} {c:12s} n={len(v):3d} mean={sum(v)/len(v):.6g}"
        print(line); o.write(line + "\n")
PY
