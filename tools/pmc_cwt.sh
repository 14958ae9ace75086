#!/bin/bash
# PMC passes for the CWT bench (tools/bench_cwt.py); each pass its own rocprofv3 run, no trace domains mixed in.
# usage: tools/pmc_cwt.sh <tag>   -> gpurun_out/pmc_<tag>/summary.txt (per kernel instantiation, mean per dispatch)
set -o pipefail
TAG=$1; shift
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/pmc_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
PASSES=(
 "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU"
 "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU"
 "SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM SQ_WAVES GRBM_GUI_ACTIVE SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR"
 "FETCH_SIZE"
 "WRITE_SIZE"
)
i=0
for P in "${PASSES[@]}"; do
  i=$((i+1))
  rocprofv3 --pmc $P --output-format csv -d $OUT/pass$i -- python3 $ROOT/tools/bench_cwt.py --steps 1 "$@" > $OUT/pass$i.log 2>&1 || echo "pass $i failed"
done
python3 - <<PY
import csv, glob, collections, re
agg = collections.defaultdict(list)
for f in glob.glob("$OUT/pass*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "cwt_" in r["Kernel_Name"]:
            name = re.sub(r"void ssq::|\(ssq::.*", "", r["Kernel_Name"])
            agg[(name + " grid=" + r.get("Grid_Size", "?"), r["Counter_Name"])].append(float(r["Counter_Value"]))
with open("$OUT/summary.txt", "w") as o:
    for (k, c), v in sorted(agg.items()):
        line = f"{k:55s} {c:26s} n={len(v):4d} mean={sum(v)/len(v):.6g}"
        print(line); o.write(line + "\n")
PY
