#!/usr/bin/env python3
"""Busy / idle accounting of one rocprofv3 --kernel-trace run: how much of the timed span has a kernel running,
how much of it is spent in tiny launches, and how large the gaps between consecutive kernels are.
usage: trace_gaps.py <dir containing *kernel_trace.csv> [n_last_steps_fraction]"""
import csv
import glob
import json
import sys

files = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)
rows = []
for f in files:
    with open(f) as fh:
        for r in csv.DictReader(fh):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
# keep the last 60 % of the trace (past model setup and warm-up)
t_lo = rows[0][0] + 0.4 * (rows[-1][1] - rows[0][0])
rows = [r for r in rows if r[0] >= t_lo]
span = rows[-1][1] - rows[0][0]
busy, cur_end = 0, rows[0][0]
gaps = []
for s, e, _ in rows:
    if s > cur_end:
        gaps.append(s - cur_end)
        busy += e - s
    else:
        busy += max(0, e - max(s, cur_end))
    cur_end = max(cur_end, e)
small = [(e - s) for s, e, _ in rows if e - s < 10_000]
out = {
    "kernels": len(rows), "span_ms": span / 1e6, "busy_ms": busy / 1e6, "idle_ms": (span - busy) / 1e6,
    "idle_frac": round(1 - busy / span, 4),
    "kernels_under_10us": len(small), "time_in_kernels_under_10us_ms": sum(small) / 1e6,
    "gaps": len(gaps), "mean_gap_us": (sum(gaps) / max(1, len(gaps))) / 1e3,
    "gaps_over_20us": sum(1 for g in gaps if g > 20_000), "time_in_gaps_over_20us_ms": sum(g for g in gaps if g > 20_000) / 1e6,
}
print(json.dumps(out, indent=1))
