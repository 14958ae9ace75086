#!/usr/bin/env python3
"""Kernel-trace statistics restricted to the TIMED dispatches of bench.py (evidence hygiene).

    rocprofv3 --kernel-trace --output-format csv -d DIR -- python3 bench.py ... > line.json
    python tools/trace_timed.py DIR line.json [kernel-name-substring]

bench.py's JSON line carries roofline.timed_dispatches = [first, last) = the launch-order indices of the interior-tile
kernel's dispatches inside the timed region (validation, run-in and warm-up launches come before them, the secondary
legs after).  Prints the average / min / max duration of exactly those dispatches and of all of them."""
import csv
import glob
import json
import os
import sys

d, line = sys.argv[1], sys.argv[2]
pat = sys.argv[3] if len(sys.argv) > 3 else "stft_tx1024_kernel<false, false, 16, false>"
j = json.loads([ln for ln in open(line).read().splitlines() if ln.startswith("{")][-1])
first, last = j["roofline"]["timed_dispatches"]
rows = []
for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        name = r.get("Kernel_Name", "")
        if pat.replace(" ", "") in name.replace(" ", ""):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"])))
rows.sort()
dur = [(e - s) / 1e6 for s, e in rows]
sel = dur[first:last]
out = {"kernel": pat, "dispatches_total": len(dur), "timed_dispatches": [first, last],
       "timed_ms_avg": sum(sel) / len(sel), "timed_ms_min": min(sel), "timed_ms_max": max(sel),
       "all_ms_avg": sum(dur) / len(dur), "all_ms_max": max(dur),
       "bench_kernel_ms_avg (HIP events, interior + edge launch)": j["roofline"]["kernel_ms_avg"],
       "bench_ms_per_step": j["ms_per_step"]}
print(json.dumps(out, indent=1))
