#!/usr/bin/env python3
"""MFMA utilisation per kernel from a rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CU_CYCLES pass.

MfmaUtil (rocprofv3's own derived metric, counters.txt) = sum(SQ_VALU_MFMA_BUSY_CYCLES) / (max(GRBM_GUI_ACTIVE) x
SIMD_NUM): matrix-pipe busy cycles over all SIMD-cycles of the dispatch.  SQ_VALU_MFMA_BUSY_CYCLES counts cycles
(= 32 x N for v_mfma_f32_32x32x16_bf16, MI355X_MICROARCH.md); SIMD_NUM = 256 CUs x 4.  Per launch, averaged."""
import collections
import csv
import glob
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

SIMDS = 256 * 4
agg = collections.defaultdict(lambda: collections.defaultdict(float))
launches = collections.defaultdict(set)
per_dispatch = collections.defaultdict(lambda: collections.defaultdict(list))
for d in sys.argv[1:]:
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0][:70]
            per_dispatch[(k, r.get("Dispatch_Id", "0"))][r["Counter_Name"]].append(float(r["Counter_Value"]))
            launches[k].add(r.get("Dispatch_Id", "0"))
# one CSV row per counter instance (SE / XCD): SQ counters add up over instances, GRBM_GUI_ACTIVE is the wall clock of
# the dispatch on every instance -> max (rocprofv3's own MfmaUtil expression: reduce(sum) / reduce(max))
XCDS = 8
for (k, _), cs in per_dispatch.items():
    for name, vals in cs.items():
        if name.startswith("GRBM"):
            # several rows: one per instance -> max.  ONE row: rocprofv3 already summed the 8 XCD instances (the value is
            # 8x the dispatch's wall cycles: checked against the kernel-trace durations) -> divide
            agg[k][name] += max(vals) if len(vals) > 1 else vals[0] / XCDS
        else:
            agg[k][name] += sum(vals)
out = {}
for k, v in agg.items():
    if not any(s in k for s in ("igemm", "mha_", "attn_q1")):
        continue
    n = max(len(launches[k]), 1)
    busy, act = v.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0), v.get("GRBM_GUI_ACTIVE", 0.0)
    if act <= 0:
        continue
    out[k] = {"launches": n, "mfma_util_pct": round(100.0 * busy / (act * SIMDS), 2),
              "mfma_busy_cycles_per_launch": round(busy / n), "gpu_active_cycles_per_launch": round(act / n)}
from bench import source_sha  # noqa: E402
print(json.dumps({"source_sha": source_sha(), "note": "MfmaUtil = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE x 1024 SIMDs), rocprofv3 --pmc, own pass; "
                          "GRBM_GUI_ACTIVE arrives summed over the 8 XCDs and is divided by 8",
                  "kernels": dict(sorted(out.items(), key=lambda kv: -kv[1]["mfma_busy_cycles_per_launch"] * kv[1]["launches"]))}))
