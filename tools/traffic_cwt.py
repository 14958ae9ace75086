"""Sum FETCH_SIZE / WRITE_SIZE over every dispatch of ONE ssq_cwt call from two rocprofv3 --pmc passes:
    python tools/traffic_cwt.py <pass_fetch_dir> <pass_write_dir> [n_calls]
(the passes of tools/pmc_cwt.sh; bench_cwt.py --steps 1 makes n_calls = 2: warm-up + timed).  Bytes as
MI355X_MICROARCH.md §HBM prescribes: 2*FETCH_SIZE (gfx950 reads 1/2) + WRITE_SIZE, in KB -> *1024."""
import collections
import csv
import glob
import json
import re
import sys

fdir, wdir = sys.argv[1], sys.argv[2]
calls = float(sys.argv[3]) if len(sys.argv) > 3 else 2.0
tot = collections.defaultdict(lambda: [0.0, 0.0, 0])
for d, col in ((fdir, 0), (wdir, 1)):
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] not in ("FETCH_SIZE", "WRITE_SIZE"):
                continue
            name = re.sub(r"void ssq::|\(ssq::.*|\(.*", "", r["Kernel_Name"])[:48]
            tot[name][col] += float(r["Counter_Value"])
            if col == 0:
                tot[name][2] += 1
rows = []
all_b = 0.0
for k, (fe, wr, n) in sorted(tot.items(), key=lambda kv: -(2 * kv[1][0] + kv[1][1])):
    b = (2 * fe + wr) * 1024 / calls
    all_b += b
    rows.append({"kernel": k, "dispatches_per_call": n / calls, "fetch_GB": 2 * fe * 1024 / calls / 1e9,
                 "write_GB": wr * 1024 / calls / 1e9, "GB": b / 1e9})
alg = 4 * (1 << 20) + 8 * 256 * (1 << 20)
print(json.dumps({"total_GB_per_call": all_b / 1e9, "algorithmic_GB_C4": alg / 1e9, "ratio": all_b / alg, "per_kernel": rows}, indent=1))
