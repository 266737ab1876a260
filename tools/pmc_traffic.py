#!/usr/bin/env python3
"""Per-kernel HBM traffic per launch from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE).
MI355X_MICROARCH.md §HBM: both counters are in KiB (x1024); on gfx950 FETCH_SIZE under-reports wide coalesced
reads by exactly 2x, so reads are doubled; WRITE_SIZE is taken as is (uncalibrated)."""
import collections
import csv
import glob
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def collect(d, counter):
    agg = collections.defaultdict(lambda: [0.0, set()])
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] != counter:
                continue
            k = r["Kernel_Name"]
            agg[k][0] += float(r["Counter_Value"])
            agg[k][1].add(r.get("Dispatch_Id", "0"))
    return {k: (v[0], len(v[1])) for k, v in agg.items()}


fetch = collect(sys.argv[1], "FETCH_SIZE")
write = collect(sys.argv[2], "WRITE_SIZE")
out = {}
for k in sorted(set(fetch) | set(write), key=lambda k: -(fetch.get(k, (0, 1))[0])):
    fs, fn = fetch.get(k, (0.0, 1))
    ws, wn = write.get(k, (0.0, 1))
    name = k.split("(")[0][:60]
    out[name] = {"launches": fn, "read_MB_per_launch": round(2 * fs * 1024 / max(fn, 1) / 1e6, 3),
                 "write_MB_per_launch": round(ws * 1024 / max(wn, 1) / 1e6, 3)}
from bench import source_sha  # noqa: E402
# steps of the traced command: one loss-forward launch per step.  (Round 5 counted stem launches; with the pipelined prefix a
# process launches steps + 1 stems — the first step computes its own prefix and stages the next one's, the last step stages
# one that nobody consumes — and the per-step figures came out a sixth too small in a 5-step pass.)
steps = sum(v["launches"] for k, v in out.items() if "stg_loss_fwd_kernel" in k)
if not steps:
    steps = sum(v["launches"] for k, v in out.items() if "igemm_stem_kernel" in k or "stem_pl_kernel" in k)
print(json.dumps({"source_sha": source_sha(), "steps_traced": steps, "note": "FETCH_SIZE x1024 x2 (gfx950 correction), WRITE_SIZE x1024; per launch", "kernels": out}))
