#!/bin/bash
# PMC passes for bench.py's dominant kernel (each pass its own rocprofv3 run; no trace domains mixed in).
# usage: tools/pmc.sh <tag> [bench args...]   -> gpurun_out/pmc_<tag>/passN/...
set -o pipefail
TAG=$1; shift
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/pmc_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
PASSES=(
 "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU"
 "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR"
 "SQ_INSTS_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM SQ_WAVES GRBM_GUI_ACTIVE"
 "FETCH_SIZE"
 "WRITE_SIZE"
)
i=0
for P in "${PASSES[@]}"; do
  i=$((i+1))
  rocprofv3 --pmc $P --output-format csv -d $OUT/pass$i -- python3 $ROOT/bench.py --no-cpu-baseline --no-secondary --no-validate --distinct 16 "$@" > $OUT/pass$i.log 2>&1 || echo "pass $i failed"
done
python3 - <<PY
import csv, glob, collections
agg = collections.defaultdict(list)
for f in glob.glob("$OUT/pass*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if any(k in r["Kernel_Name"] for k in ("stft_fused", "stft_tx1024", "cwt_")):
            agg[(r["Kernel_Name"][:60], r["Counter_Name"])].append(float(r["Counter_Value"]))
with open("$OUT/summary.txt", "w") as o:
    for (k, c), v in sorted(agg.items()):
        line = f"{k:60s} {c:28s} n={len(v):3d} mean={sum(v)/len(v):.6g}"
        print(line); o.write(line + "\n")
PY
