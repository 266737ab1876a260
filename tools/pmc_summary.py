#!/usr/bin/env python3
"""Average rocprofv3 --pmc counter values per kernel (per launch) from *counter_collection.csv files."""
import collections
import csv
import glob
import sys

for d in sys.argv[1:]:
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        agg = collections.defaultdict(lambda: collections.defaultdict(float))
        launches = collections.defaultdict(set)
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"][:70]
            agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
            launches[k].add(r.get("Dispatch_Id", r.get("Correlation_Id", "0")))
        for k, v in agg.items():
            if "igemm" in k or "mha" in k:
                n = max(len(launches[k]), 1)
                print(k, {c: round(x / n) for c, x in v.items()}, "launches", n)
